// The reference's ray-primitive tests, operation for operation: node bounding sphere
// (Sphere.h:155-204), triangle (Triangle.h:91-158), sphere (Sphere.h:35-77).
#pragma once
#include "device_math.hpp"
#include "device_records.hpp"

// Bounding-sphere test of a node (Sphere.h:155-204).  Returns the value the reference
// returns (t2, t1 or `sentinel`).  The reference takes sqrt(B*B-4AC) without halving it, so
// its roots are wider than the sphere's; *entry receives the true distance at which the
// ray enters the sphere, (-b - root/2)/A, which is what pruning may compare against.
DEV double bound_test(const rt_node *nd, d3 o, d3 d, double A, double minT, double sentinel, double *entry)
{
    d3 emp = sub3(o, ld3(nd->center));
    double r = nd->radius;
    double C = dot3(emp, emp) - r * r;
    double B = 2.0 * dot3(d, emp);
    double disc = B * B - 4 * A * C;
    *entry = sentinel;
    if (disc < 0) return sentinel;
    double b = B / 2;
    double root = sqrt(disc);
    double t1 = (-b + root) / A;
    double t2 = (-b - root) / A;
    *entry = (-b - 0.5 * root) / A;
    if (t2 > minT) return t2;
    else if (t1 > minT) return t1;
    return sentinel;
}

// Triangle::intersect (Triangle.h:91-158) without the hit-record update; returns t or maxT.
DEV double tri_test(const rt_tri_rec *tr, d3 o, d3 d, double minT, double maxT, double *alpha_out, double *beta_out)
{
    d3 v0 = ld3(tr->v0), ab = ld3(tr->ab), ac = ld3(tr->ac);
    d3 ae = sub3(v0, o);
    double detA = det3(ab, ac, d);
    double alpha = det3(ae, ac, d) / detA;
    if (alpha < 0 || alpha > 1) return maxT;
    double beta = det3(ab, ae, d) / detA;
    if (beta < 0 || beta > 1 - alpha) return maxT;
    double t = det3(ab, ac, ae) / detA;
    if (t < minT || t > maxT) return maxT;
    *alpha_out = alpha;
    *beta_out = beta;
    return t;
}

// Operands of one triangle test, fetched as five 16-byte loads (the first 80 bytes of the record:
// v0, ab, ac, primitive id, tie key).  The memory pipeline spends the same time on a one-word load
// of a wave as on a four-word one, so the number of load instructions is what counts.
struct TriOps {
    d3 v0, ab, ac;
    int prim, key;
};
DEV TriOps load_tri(const rt_tri_rec *tr)
{
    typedef double v2d __attribute__((ext_vector_type(2)));
    const __attribute__((address_space(1))) v2d *p = (const __attribute__((address_space(1))) v2d *)tr;
    const v2d a = p[0], b = p[1], c = p[2], e = p[3], f = p[4];
    TriOps t;
    t.v0 = mk3(a.x, a.y, b.x);
    t.ab = mk3(b.y, c.x, c.y);
    t.ac = mk3(e.x, e.y, f.x);
    t.prim = __double2loint(f.y);
    t.key = __double2hiint(f.y);
    return t;
}

// The same test with its decisions taken before any division wherever that is safe.  The four
// determinants are built from shared 2x2 minors in the reference's association, so detA, X, Y, Z are
// bit-identical to det3().  With s = sign(detA) and ad = |detA| (both in a range where no quotient
// over- or underflows), rounding to nearest is monotonic, so for the quotients the reference forms:
//   alpha = fl(X/detA) < 0  exactly when sX < 0 (unless the quotient underflows);  alpha > 1  is false for sure
//   when sX <= ad and true for sure when sX > ad (1 + 2^-40);  likewise beta < 0;
//   beta > 1 - alpha  is false for sure when sX + sY <= ad (1 - 2^-30) and true for sure when
//   sX + sY > ad (1 + 2^-30);   t against minT, maxT: for sure outside a factor 1 +- 2^-40 of the threshold.
// A miss is declared and a hit's barycentric tests are passed on these grounds; only a lane that falls
// between the sure cases (or out of range) runs the reference's exact sequence of three divisions.
// TriGate: what the sure cases say about one lane.
struct TriGate {
    double detA, Z, ad, sZ;
    bool miss;       // no intersection, for sure
    bool inside;     // alpha and beta pass the reference's tests, for sure (t still to be judged)
    bool early;      // the first barycentric coordinate ruled the triangle out for every lane testing one
};
// v * sign(detA), by the sign bit (sgn = the sign bit of detA's high word): -0 and NaN pass through as they should
DEV double tri_flip(double v, uint32_t sgn) { return __hiloint2double((int)((uint32_t)__double2hiint(v) ^ sgn), __double2loint(v)); }
DEV TriGate tri_gate(const TriOps &tr, d3 o, d3 d, double *Xo, double *Yo)
{
    TriGate g;
    const d3 v0 = tr.v0, ab = tr.ab, ac = tr.ac;
    const double m3 = ab.y * ac.z - ac.y * ab.z;
    const d3 ae = sub3(v0, o);
    const double m1 = ac.y * d.z - d.y * ac.z;
    const double m2 = d.y * ab.z - ab.y * d.z;
    const double detA = ab.x * m1 + ac.x * m2 + d.x * m3;
    const double m4 = d.y * ae.z - ae.y * d.z;
    const double m5 = ae.y * ac.z - ac.y * ae.z;
    const double X = ae.x * m1 + ac.x * m4 + d.x * m5;          // det3(ae, ac, d)
    const uint32_t sgn = (uint32_t)__double2hiint(detA) & 0x80000000u;
    const double ad = fabs(detA);
    // |detA| in a range where a quotient by it neither overflows to a wrong side of a threshold nor underflows to zero: a
    // numerator below -1e-150 then gives a quotient below zero, an infinite one compares like the infinite quotient would,
    // and a NaN fails every comparison below, which sends the lane to the exact sequence
    const bool ranged = ad > 1e-150 && ad < 1e150;
    const double sX = tri_flip(X, sgn);
    const double up = ad * (1.0 + 0x1p-40);
    g.detA = detA; g.ad = ad; g.Z = 0; g.sZ = 0; g.inside = false;
    *Xo = X; *Yo = 0;
    // When the first barycentric coordinate alone rules the triangle out for every lane of the wave that is
    // testing one, the other two determinants are not formed (neighbouring rays mostly agree on that).
    const bool miss_x = ranged && (sX < -1e-150 || sX > up);
    g.miss = miss_x;
    g.early = __ballot(!miss_x) == 0;
    if (g.early) return g;
    const double m6 = ab.y * ae.z - ae.y * ab.z;
    const double Y = ab.x * (-m4) + ae.x * m2 + d.x * m6;       // det3(ab, ae, d)
    const double Z = ab.x * (-m5) + ac.x * (-m6) + ae.x * m3;   // det3(ab, ac, ae)
    const double sY = tri_flip(Y, sgn);
    const double sum = sX + sY;
    g.miss = miss_x || (ranged && (sY < -1e-150 || sum > ad * (1.0 + 0x1p-30)));
    g.inside = ranged && sX >= 0 && sX <= ad && sY >= 0 && sum <= ad * (1.0 - 0x1p-30);
    g.Z = Z; g.sZ = tri_flip(Z, sgn);
    *Yo = Y;
    return g;
}
DEV double tri_exact_tail(double X, double Y, double Z, double detA, double minT, double maxT)
{
    double alpha = X / detA;
    if (alpha < 0 || alpha > 1) return maxT;
    double beta = Y / detA;
    if (beta < 0 || beta > 1 - alpha) return maxT;
    double t = Z / detA;
    if (t < minT || t > maxT) return maxT;
    return t;
}
// Nearest-hit form: returns t or maxT.  `tcut` <= maxT: a hit farther than tcut may be reported as
// maxT (it cannot replace the current nearest hit).
DEV double tri_test_fast(const TriOps &tr, d3 o, d3 d, double minT, double maxT, double tcut)
{
    double X, Y;
    const TriGate g = tri_gate(tr, o, d, &X, &Y);
    if (g.early) return maxT;
    // a hit for sure farther than tcut, or nearer than minT, needs no quotient either
    const bool wanted = g.inside && !(g.sZ < g.ad * (minT * (1.0 - 0x1p-40)) || g.sZ > (tcut * g.ad) * (1.0 + 0x1p-40));
    double t = maxT;
    const bool exact = !(g.miss || g.inside);
    if (__ballot(wanted)) {
        if (wanted) {
            t = g.Z / g.detA;
            if (t < minT || t > maxT) t = maxT;
        }
    }
    if (__ballot(exact)) {
        if (exact) t = tri_exact_tail(X, Y, g.Z, g.detA, minT, maxT);
    }
    return t;
}
// Any-hit form: is there an intersection with t in [minT, maxT]?  No division unless a lane falls between
// the sure cases.
DEV bool tri_any_fast(const TriOps &tr, d3 o, d3 d, double minT, double maxT)
{
    double X, Y;
    const TriGate g = tri_gate(tr, o, d, &X, &Y);
    if (g.early) return false;
    // (the margins of 2^-40 dwarf the roundings of these products, whichever way they are associated)
    const double hi = maxT * g.ad;
    const bool t_out = g.sZ < g.ad * (minT * (1.0 - 0x1p-40)) || g.sZ > hi * (1.0 + 0x1p-40);
    const bool t_in = g.sZ >= g.ad * (minT * (1.0 + 0x1p-40)) && g.sZ <= hi * (1.0 - 0x1p-40);
    bool hit = g.inside && t_in;
    const bool exact = !(g.miss || (g.inside && (t_in || t_out)));
    if (__ballot(exact)) {
        if (exact) hit = tri_exact_tail(X, Y, g.Z, g.detA, minT, maxT) < maxT;
    }
    return hit;
}

// Sphere::intersect (Sphere.h:35-77) without the hit-record update; returns t or maxT.
DEV double sph_test(const rt_sph_rec *sp, d3 o, d3 d, double d2, double minT, double maxT)
{
    d3 emc = sub3(o, ld3(sp->center));
    double r = sp->radius;
    double ddotemc = dot3(d, emc);
    double disc = (ddotemc * ddotemc) - d2 * (dot3(emc, emc) - r * r);
    if (disc < 0) return maxT;
    double sq = sqrt(disc);
    double root1 = (-ddotemc + sq) / d2;
    double root2 = (-ddotemc - sq) / d2;
    double t = root1;
    if (root1 < 0 || (root2 > minT && root2 < root1)) t = root2;
    if (t < minT || t > maxT) return maxT;
    return t;
}

