// 3-vector arithmetic in the evaluation order of the reference's vector library (Eigen 3.4 fixed-size
// expressions): dot = (x0*y0 + x1*y1) + x2*y2, normalized = v / sqrt(v.v) when v.v > 0.  Compiled with
// -ffp-contract=off, every operation is the IEEE operation the CPU performs.
#pragma once
#include <hip/hip_runtime.h>

#define DEV __device__ __forceinline__
#define DEV_HOST_INLINE __host__ __device__ __forceinline__

struct d3 { double x, y, z; };
DEV d3 mk3(double x, double y, double z) { d3 r; r.x = x; r.y = y; r.z = z; return r; }
DEV d3 ld3(const double *p) { return mk3(p[0], p[1], p[2]); }
DEV d3 sub3(d3 a, d3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
DEV d3 add3(d3 a, d3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
DEV d3 neg3(d3 a) { return mk3(-a.x, -a.y, -a.z); }
DEV d3 scale3(double s, d3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
DEV d3 div3(d3 a, double s) { return mk3(a.x / s, a.y / s, a.z / s); }
DEV double dot3(d3 a, d3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
DEV d3 unit3(d3 a)
{
    double z = dot3(a, a);
    if (z > 0) { double s = sqrt(z); return div3(a, s); }
    return a;
}
DEV double norm3(d3 a) { return sqrt(dot3(a, a)); }

// det of the matrix with columns a, b, c in the reference's association (Triangle.h:388-393)
DEV double det3(d3 a, d3 b, d3 c)
{
    return a.x * (b.y * c.z - c.y * b.z) + b.x * (c.y * a.z - a.y * c.z) + c.x * (a.y * b.z - b.y * a.z);
}

