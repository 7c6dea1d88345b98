// What happens at a hit: the hit record's geometry (Solid.h:110-125), refraction (Camera.h:260-302),
// the Phong terms of one light (Solid.h:128-172), and the counter-based stand-in for rand().
#pragma once
#include "primitives.hpp"

// Geometry of a found hit (Solid.h:110-125): primitive id, unit normal, impact point.
// `best` is a slot of the given record arrays (| SLOT_SPHERE for spheres).
DEV void hit_geometry(const DevScene &S, const rt_tri_rec *tri_arr, const rt_sph_rec *sph_arr, const uint32_t *sph_prim_arr,
                      uint32_t flags, d3 o, d3 d, double minT, double t, int best, int known_prim, int *prim_out, d3 *n_out,
                      d3 *P_out, bool *flat_out = nullptr)
{
    d3 nraw;
    bool flat = false;
    int prim = known_prim;       // >= 0: the walk already knows it, no need to go through the record
    if (best & SLOT_SPHERE) {
        const int s = best & ~SLOT_SPHERE;
        if (known_prim < 0) prim = (int)(sph_prim_arr[s] & ~RT_SPH_TRANSPARENT);
        const rt_sph_rec *sp = &sph_arr[s];
        nraw = div3(sub3(add3(o, scale3(t, d)), ld3(sp->center)), sp->radius);
    } else {
        const rt_tri_rec *tr = &tri_arr[best];
        if (known_prim < 0) prim = tr->prim;
        const double *pn = &S.prim_normals[(size_t)prim * 12];
        nraw = ld3(pn);
        flat = true;
        if ((flags & RT_FLAG_PHONG_NORMALS) && (S.prim_flags[prim] & RT_PRIM_HAS_VNORMALS)) {
            flat = false;
            double al, be;
            (void)tri_test(tr, o, d, minT, 10000000000.0, &al, &be);
            d3 n0 = ld3(pn + 3), n1 = ld3(pn + 6), n2 = ld3(pn + 9);
            nraw = add3(add3(scale3(al, n1), scale3(be, n2)), scale3(1 - (al + be), n0));
        }
    }
    *prim_out = prim;
    // a flat triangle's normal is the same at every hit: its normalisations are tabulated at upload
    if (flat && flat_out) { *n_out = ld3(&S.prim_units[(size_t)prim * 9]); *flat_out = true; }
    else { *n_out = unit3(nraw); if (flat_out) *flat_out = false; }
    *P_out = mk3(o.x + d.x * t, o.y + d.y * t, o.z + d.z * t);
}

// Refraction / total-internal-reflection direction at a transparent hit (Camera.h:260-302,
// LightCam.h:397-472).  `renormalize` is the light camera's variant (LightCam.h:440).
DEV d3 refract_dir(d3 d, d3 n, double ior, bool renormalize)
{
    d3 nn = n;
    d3 V = neg3(d);
    double dt = dot3(nn, V);
    if (dt >= 0) ior = 1.0 / ior;
    else { nn = neg3(nn); dt = -dt; }
    double disc = 1 - (1 - (dt * dt)) * (ior * ior);
    if (disc > 0) {
        double sq = sqrt(disc);
        d3 rd = mk3((nn.x * dt - V.x) * ior - nn.x * sq, (nn.y * dt - V.y) * ior - nn.y * sq,
                    (nn.z * dt - V.z) * ior - nn.z * sq);
        return renormalize ? unit3(rd) : rd;
    }
    return unit3(add3(d, scale3(2.0 * -dot3(d, nn), nn)));
}

// The power of the specular term, pow(a, s) for a base that is not negative (a cosine, clamped by the caller), as
// exp(s log a) from a logarithm and an exponential written for this use (the classic scheme of fdlibm's e_log.c and e_exp.c:
// argument reduction by the binary exponent, a rational argument and an even polynomial; each below one ulp of error): some
// 90 vector instructions and few registers where the device library's pow() takes 224 instructions and more registers than
// anything else in the kernels that shade.  The relative error is at most (3 |s log a| + 2) 2^-53 (the logarithm's ulp and the
// product's rounding, magnified by the exponential, and the exponential's own ulp) -- where the term can be seen at all
// (s log a > -36) that is 1.3e-14 of it, and the device library's pow() and the host's differ in the last place already.  pow()'s special values for bases >= 0 are kept: base 0 or infinite, exponent 0, infinite or not a number, base 1.
DEV double log_of_cosine(double x)
{
    double m = __builtin_amdgcn_frexp_mant(x);          // x = 2^k m, m in [1/2, 1); subnormals included
    int k = __builtin_amdgcn_frexp_exp(x);
    if (m < 0.70710678118654752440) { m = m + m; k -= 1; }
    const double f = m - 1.0;                            // exact: m in [sqrt(1/2), sqrt(2))
    const double q = f / (2.0 + f);
    const double z = q * q, w = z * z;
    const double t1 = w * __builtin_fma(w, __builtin_fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
    const double t2 = z * __builtin_fma(w, __builtin_fma(w, __builtin_fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01),
                                        6.666666666666735130e-01);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)k;
    // (k ln2 in two parts: the high one has 32 trailing zero bits, so its product with k is exact)
    const double r = dk * 6.93147180369123816490e-01 - ((hfsq - (q * (hfsq + R) + dk * 1.90821492927058770002e-10)) - f);
    if (!(x < __builtin_inf())) return x;               // (infinite, or not a number)
    return x == 0.0 ? -__builtin_inf() : r;
}
DEV double exp_of_product(double y)
{
    const double kd = __builtin_rint(y * 1.44269504088896338700e+00);
    const double hi = __builtin_fma(-kd, 6.93147180369123816490e-01, y);
    const double lo = kd * 1.90821492927058770002e-10;
    const double r = hi - lo;
    const double t = r * r;
    const double c = r - t * __builtin_fma(t, __builtin_fma(t, __builtin_fma(t, __builtin_fma(t, 4.13813679705723846039e-08, -1.65339022054652515390e-06),
                                                                                6.61375632143793436117e-05), -2.77777777770155933842e-03), 1.66666666666666019037e-01);
    const double e = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
    double v = __builtin_amdgcn_ldexp(e, (int)kd);       // (rounds once, also into the subnormals)
    if (y < -745.2) v = 0.0;
    if (y > 709.8) v = __builtin_inf();
    return v;
}
DEV double spec_pow(double a, double s)
{
#ifdef RT_LIBRARY_POW
    return pow(a, s);
#endif
    const double v = exp_of_product(s * log_of_cosine(a));
    return (s == 0.0 || a == 1.0) ? 1.0 : v;       // (pow(x, 0) = pow(1, y) = 1 whatever the other operand)
}

// Diffuse + specular terms of one light at one hit (Solid.h:128-172), scaled by `kfac`
// (Camera.h:465-476).  `rd` is the unit vector from the hit point to the light.
// `normal`, `nn`, `v`: the hit record's n, nn and v.
DEV d3 phong_terms(const rt_material &m, d3 normal, d3 nn, d3 v, const rt_light *lt, d3 rd, double kfac)
{
    d3 local = mk3(0, 0, 0);
    if (m.Ks > 0) {
        d3 rdir = unit3(add3(neg3(rd), scale3(2.0 * dot3(rd, nn), nn)));
        double a = dot3(rdir, v);
        // pow(max(a, 0), Shine) (Solid.h:166-169).  Behind the mirror direction the base is +0 and the
        // value follows from the exponent alone; pow() itself costs as much as the rest of this function
        // and neighbouring lanes are mostly on the same side, so whole waves skip it.
        double pw;
        if (a < 0.0) pw = m.Shine > 0.0 ? 0.0 : (m.Shine == 0.0 ? 1.0 : (m.Shine < 0.0 ? __builtin_inf() : m.Shine));
        else pw = spec_pow(a, m.Shine);
        double ik = lt->intensity * m.Ks;
        d3 comp = mk3((ik * lt->color[0]) * pw, (ik * lt->color[1]) * pw, (ik * lt->color[2]) * pw);
        local = add3(local, scale3(kfac, comp));
    }
    {
        double a = dot3(rd, normal);
        double dm = (a < 0.0) ? 0.0 : a;
        double ik = lt->intensity * m.Kd;
        d3 comp = mk3((ik * m.r) * dm, (ik * m.g) * dm, (ik * m.b) * dm);
        local = add3(local, scale3(kfac, comp));
    }
    return local;
}

// True when phong_terms() of this light at this hit is zero in every component for sure, whatever the factor: the
// diffuse term is `coefficients * max(rd.normal, 0)` and the specular term `coefficients * pow(max(rdir.v, 0), Shine)`
// (Solid.h:133,166-167), so with bounded coefficients a light from behind the shading normal whose mirror direction
// points away from the viewer adds +-0 to the colour, seen or not (Camera.h:465-476).  The mirror direction is judged
// before its normalisation with a margin far beyond the rounding of the normalised form: |raw| <= 3 and |v| <= 1, so
// raw.v < -1e-6 puts the normalised product below -3e-7 where its rounding error is about 1e-15.
DEV bool light_adds_nothing(double Kd, double Ks, double Shine, d3 normal, d3 nn, d3 v, d3 rd)
{
    const double a = dot3(rd, normal);
    if (!(a < 0.0 || (Kd == 0.0 && fabs(a) < 1e100))) return false;
    if (!(Ks > 0)) return true;
    if (!(Shine > 0.0)) return false;
    const d3 raw = add3(neg3(rd), scale3(2.0 * dot3(rd, nn), nn));
    return dot3(raw, v) < -1e-6 && dot3(nn, nn) < 1.001 && dot3(v, v) < 1.001;
}

// Counter-based stand-in for the reference's shared rand() % 1000 / 1000 (Camera.h:99-103).
DEV uint32_t hash_rng(uint32_t pixel, uint32_t sample, uint32_t dim, uint32_t seed)
{
    uint32_t h = seed * 0x9E3779B1u + 0x85EBCA6Bu;
    h ^= pixel + 0x9E3779B9u + (h << 6) + (h >> 2);
    h ^= sample + 0x9E3779B9u + (h << 6) + (h >> 2);
    h ^= dim + 0x9E3779B9u + (h << 6) + (h >> 2);
    h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
    return h;
}
// k / 1000.0 for the integers k = 0..999 without the division: q = RN(k * c) with c = RN(1/1000), the residual k - 1000 q
// is exact in a fused multiply-add, and RN(q + residual * c) is the correctly rounded quotient -- for every one of the
// thousand values, which rt_create checks one by one against the host's division (rand_quotients_exact in rt_device.hip).
DEV_HOST_INLINE double thousandth(double k)
{
    const double q = k * 0.001;
    const double r = __builtin_fma(-1000.0, q, k);
    return __builtin_fma(r, 0.001, q);
}
DEV double normed_rand(uint32_t pixel, uint32_t sample, uint32_t dim, uint32_t seed)
{
    return thousandth((double)(hash_rng(pixel, sample, dim, seed) % 1000u));
}

