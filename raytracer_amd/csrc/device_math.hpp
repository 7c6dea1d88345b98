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
// a / s for s = sqrt(z) > 0: the three IEEE quotients, with the work on the divisor done once.  The compiler's division
// (scale both operands, reciprocal, two Newton steps, quotient, exact remainder, one correction, fix-up: eleven instructions)
// refines the divisor's reciprocal anew for every numerator; here one refined reciprocal serves all three, and each
// quotient is the same product, exact remainder and correction.  Without the scaling that sequence rounds correctly as long
// as nothing in it leaves the normal range, which holds when the divisor is a square root of a finite number (2^-537 ..
// 2^512) and a numerator is zero or at least 2^-450 in magnitude (it cannot exceed the root by more than a rounding); the
// sign of a zero numerator is put back by hand.  A wave in which some lane holds anything else -- a tiny or odd component,
// an overflowed square -- takes the compiler's division for those lanes.
DEV d3 div3_by_root(d3 a, double s, double z)
{
    const double r0 = __builtin_amdgcn_rcp(s);
    const double e0 = __builtin_fma(-s, r0, 1.0);
    const double r1 = __builtin_fma(e0, r0, r0);
    const double e1 = __builtin_fma(-s, r1, 1.0);
    const double r = __builtin_fma(e1, r1, r1);
    double qx = a.x * r, qy = a.y * r, qz = a.z * r;
    qx = __builtin_fma(__builtin_fma(-s, qx, a.x), r, qx);
    qy = __builtin_fma(__builtin_fma(-s, qy, a.y), r, qy);
    qz = __builtin_fma(__builtin_fma(-s, qz, a.z), r, qz);
    d3 q = mk3(__builtin_copysign(qx, a.x), __builtin_copysign(qy, a.y), __builtin_copysign(qz, a.z));
    const double least = __builtin_fmin(__builtin_fmin(__builtin_fabs(a.x), __builtin_fabs(a.y)), __builtin_fabs(a.z));
    const bool look = !(least >= 0x1p-450) || !(z < 0x1p1000);
    if (__ballot(look)) {
        // (mostly a component that is exactly zero, which the quick form handles)
        const bool odd = !(z < 0x1p1000) || (!(__builtin_fabs(a.x) >= 0x1p-450) && a.x != 0.0) || (!(__builtin_fabs(a.y) >= 0x1p-450) && a.y != 0.0) ||
                         (!(__builtin_fabs(a.z) >= 0x1p-450) && a.z != 0.0);
        if (odd) q = div3(a, s);
    }
    return q;
}
DEV d3 unit3(d3 a)
{
    double z = dot3(a, a);
    if (z > 0) { double s = sqrt(z); return div3_by_root(a, s, z); }
    return a;
}
DEV double norm3(d3 a) { return sqrt(dot3(a, a)); }

// det of the matrix with columns a, b, c in the reference's association (Triangle.h:388-393)
DEV double det3(d3 a, d3 b, d3 c)
{
    return a.x * (b.y * c.z - c.y * b.z) + b.x * (c.y * a.z - a.y * c.z) + c.x * (a.y * b.z - b.y * a.z);
}

