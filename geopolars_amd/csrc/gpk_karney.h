// gpk_karney.h — Karney's inverse geodesic on WGS84 for `geodesic_length(method = "geodesic")`, the default of the reference's
// Python surface (georust/geoseries.py:128-146; py-geopolars/src/geo.rs:61-78; geoseries.rs:52-58): geo 0.27's GeodesicLength
// sums geographiclib-rs' Geodesic::inverse over the segments of a linestring.  Restated from the published algorithm (C. F. F.
// Karney, "Algorithms for geodesics", J. Geodesy 87, 2013, and GeographicLib's order-6 series): reduced latitudes, a starting
// azimuth (spherical, or the astroid solution near the antipode), Newton's method on lambda12(alp1) with bisection as the
// fallback, then the distance integral by Clenshaw summation.  One lane = one segment; ~1 microsecond of f64 work.
#pragma once

#include <hip/hip_runtime.h>

namespace gpk {
namespace karney {

#define KQ __device__ inline
#define K_A 6378137.0
#define K_F (1.0 / 298.257223563)
#define K_F1 (1.0 - K_F)
#define K_E2 (K_F * (2.0 - K_F))
#define K_EP2 (K_E2 / (K_F1 * K_F1))
#define K_N (K_F / (2.0 - K_F))
#define K_B (K_A * K_F1)
#define K_PI 3.14159265358979323846
#define K_DEGREE (K_PI / 180.0)
#define K_TINY 1.4916681462400413e-154 /* sqrt(DBL_MIN) */
#define K_TOL0 2.220446049250313e-16   /* DBL_EPSILON */
#define K_TOL1 (200.0 * K_TOL0)
#define K_TOL2 1.4901161193847656e-08  /* sqrt(DBL_EPSILON) */
#define K_TOLB (K_TOL0 * K_TOL2)
#define K_XTHRESH (1000.0 * K_TOL2)
#define K_ETOL2 (0.1 * K_TOL2 / sqrt(fmax(0.001, fabs(K_F)) * fmin(1.0, 1.0 - K_F / 2.0) / 2.0))
#define K_MAXIT1 20
#define K_MAXIT2 83
/* ---- Karney's inverse geodesic on WGS84 (C. F. F. Karney, "Algorithms for geodesics", J. Geodesy 87, 2013; the order-6
 * series and the Newton / bisection scheme of GeographicLib's Geodesic::Inverse, which geo 0.27 reaches through the
 * geographiclib-rs crate: GeodesicDistance / GeodesicLength).  Only the distance s12 is produced. ------------------------------ */
KQ double k_sq(double x) { return x * x; }
KQ void k_norm2(double* s, double* c) {
    const double r = hypot(*s, *c);
    *s /= r;
    *c /= r;
}
KQ double k_sumx(double u, double v, double* t) {
    const double s = u + v;
    double up = s - v, vpp = s - up;
    up -= u;
    vpp -= v;
    *t = s != 0.0 ? 0.0 - (up + vpp) : s;
    return s;
}
KQ double k_ang_round(double x) {
    const double z = 1.0 / 16.0;
    double y = fabs(x);
    y = y < z ? z - (z - y) : y;
    return copysign(y, x);
}
KQ double k_ang_diff(double x, double y, double* e) {
    double t;
    double d = k_sumx(remainder(-x, 360.0), remainder(y, 360.0), &t);
    d = k_sumx(remainder(d, 360.0), t, &t);
    if (d == 0.0 || fabs(d) == 180.0) d = copysign(d, t == 0.0 ? y - x : -t);
    *e = t;
    return d;
}
KQ void k_sincosd(double x, double* sinx, double* cosx) {
    int q = 0;
    double r = remquo(x, 90.0, &q);
    r *= K_DEGREE;
    const double s = sin(r), c = cos(r);
    double sx, cx;
    switch ((unsigned)q & 3u) {
        case 0u: sx = s; cx = c; break;
        case 1u: sx = c; cx = -s; break;
        case 2u: sx = -s; cx = -c; break;
        default: sx = -c; cx = s; break;
    }
    cx += 0.0;
    if (sx == 0.0) sx = copysign(sx, x);
    *sinx = sx;
    *cosx = cx;
}
/* sum of c[k] sin(2 k x), k = 1 .. n (Clenshaw) */
KQ double k_sin_series(double sinx, double cosx, const double* c, int n) {
    const double ar = 2.0 * (cosx - sinx) * (cosx + sinx);
    c += n + 1;
    double y0 = (n & 1) ? *--c : 0.0, y1 = 0.0;
    n /= 2;
    while (n--) {
        y1 = ar * y0 - y1 + *--c;
        y0 = ar * y1 - y0 + *--c;
    }
    return 2.0 * sinx * cosx * y0;
}
KQ double k_A1m1f(double eps) {
    const double e2 = eps * eps, t = e2 * (e2 * (e2 + 4.0) + 64.0) / 256.0;
    return (t + eps) / (1.0 - eps);
}
KQ void k_C1f(double eps, double* c) {
    const double e2 = eps * eps;
    double d = eps;
    c[1] = d * ((6.0 - e2) * e2 - 16.0) / 32.0;
    d *= eps;
    c[2] = d * ((64.0 - 9.0 * e2) * e2 - 128.0) / 2048.0;
    d *= eps;
    c[3] = d * (9.0 * e2 - 16.0) / 768.0;
    d *= eps;
    c[4] = d * (3.0 * e2 - 5.0) / 512.0;
    d *= eps;
    c[5] = -7.0 * d / 1280.0;
    d *= eps;
    c[6] = -7.0 * d / 2048.0;
}
KQ double k_A2m1f(double eps) {
    const double e2 = eps * eps, t = e2 * (e2 * (-11.0 * e2 - 28.0) - 192.0) / 256.0;
    return (t - eps) / (1.0 + eps);
}
KQ void k_C2f(double eps, double* c) {
    const double e2 = eps * eps;
    double d = eps;
    c[1] = d * (e2 * (e2 + 2.0) + 16.0) / 32.0;
    d *= eps;
    c[2] = d * (e2 * (35.0 * e2 + 64.0) + 384.0) / 2048.0;
    d *= eps;
    c[3] = d * (15.0 * e2 + 80.0) / 768.0;
    d *= eps;
    c[4] = d * (7.0 * e2 + 35.0) / 512.0;
    d *= eps;
    c[5] = 63.0 * d / 1280.0;
    d *= eps;
    c[6] = 77.0 * d / 2048.0;
}
KQ double k_A3f(double eps) {
    const double n = K_N;
    double y = -3.0 / 128.0;
    y = y * eps + (-2.0 * n - 3.0) / 64.0;
    y = y * eps + ((-n - 3.0) * n - 1.0) / 16.0;
    y = y * eps + ((3.0 * n - 1.0) * n - 2.0) / 8.0;
    y = y * eps + (n - 1.0) / 2.0;
    y = y * eps + 1.0;
    return y;
}
KQ void k_C3f(double eps, double* c) {
    const double n = K_N;
    double m = eps;  /* eps^l */
    c[1] = m * ((((3.0 / 128.0) * eps + (2.0 * n + 5.0) / 128.0) * eps + ((-n + 3.0) * n + 3.0) / 64.0) * eps + ((-n) * n + 1.0) / 8.0) * eps
           + m * ((-n + 1.0) / 4.0);
    m *= eps;
    c[2] = m * ((((5.0 / 256.0) * eps + (n + 3.0) / 128.0) * eps + ((-3.0 * n - 2.0) * n + 3.0) / 64.0) * eps + ((n - 3.0) * n + 2.0) / 32.0);
    m *= eps;
    c[3] = m * (((7.0 / 512.0) * eps + (-10.0 * n + 9.0) / 384.0) * eps + ((5.0 * n - 9.0) * n + 5.0) / 192.0);
    m *= eps;
    c[4] = m * ((7.0 / 512.0) * eps + (-14.0 * n + 7.0) / 512.0);
    m *= eps;
    c[5] = m * (21.0 / 2560.0);
}
/* s12b (distance / b) and m12b (reduced length / b) of the arc sig1 .. sig2; *m0 = A1 - A2 */
KQ void k_lengths(double eps, double sig12, double ssig1, double csig1, double dn1, double ssig2, double csig2, double dn2, double* s12b, double* m12b,
               double* m0) {
    double ca[7], cb[7];
    k_C1f(eps, ca);
    k_C2f(eps, cb);
    const double A1 = k_A1m1f(eps), A2 = k_A2m1f(eps);
    const double B1 = k_sin_series(ssig2, csig2, ca, 6) - k_sin_series(ssig1, csig1, ca, 6);
    const double B2 = k_sin_series(ssig2, csig2, cb, 6) - k_sin_series(ssig1, csig1, cb, 6);
    *s12b = (1.0 + A1) * (sig12 + B1);
    *m0 = A1 - A2;
    const double J12 = (A1 - A2) * sig12 + ((1.0 + A1) * B1 - (1.0 + A2) * B2);
    *m12b = dn2 * (csig1 * ssig2) - dn1 * (ssig1 * csig2) - csig1 * csig2 * J12;
}
KQ double k_astroid(double x, double y) {
    const double p = x * x, q = y * y, r = (p + q - 1.0) / 6.0;
    if (q == 0.0 && r <= 0.0) return 0.0;
    const double S = p * q / 4.0, r2 = r * r, r3 = r * r2, disc = S * (S + 2.0 * r3);
    double u = r;
    if (disc >= 0.0) {
        double T3 = S + r3;
        T3 += T3 < 0.0 ? -sqrt(disc) : sqrt(disc);
        const double T = cbrt(T3);
        u += T + (T != 0.0 ? r2 / T : 0.0);
    } else {
        const double ang = atan2(sqrt(-disc), -(S + r3));
        u += 2.0 * r * cos(ang / 3.0);
    }
    const double v = sqrt(u * u + q), uv = u < 0.0 ? q / (v - u) : u + v, w = (uv - q) / (2.0 * v);
    return uv / (sqrt(uv + w * w) + w);
}
/* starting point of Newton's method; returns sig12 >= 0 when the line is short enough to be settled at once */
KQ double k_inverse_start(double sbet1, double cbet1, double dn1, double sbet2, double cbet2, double dn2, double lam12, double slam12, double clam12,
                       double* psalp1, double* pcalp1, double* psalp2, double* pcalp2, double* pdnm) {
    double salp1 = 0.0, calp1 = 0.0, salp2 = 0.0, calp2 = 0.0, dnm = 0.0;
    double sig12 = -1.0;
    const double sbet12 = sbet2 * cbet1 - cbet2 * sbet1, cbet12 = cbet2 * cbet1 + sbet2 * sbet1, sbet12a = sbet2 * cbet1 + cbet2 * sbet1;
    const int shortline = cbet12 >= 0.0 && sbet12 < 0.5 && cbet2 * lam12 < 0.5;
    double somg12, comg12;
    if (shortline) {
        double sbetm2 = k_sq(sbet1 + sbet2);
        sbetm2 /= sbetm2 + k_sq(cbet1 + cbet2);
        dnm = sqrt(1.0 + K_EP2 * sbetm2);
        const double omg12 = lam12 / (K_F1 * dnm);
        somg12 = sin(omg12);
        comg12 = cos(omg12);
    } else {
        somg12 = slam12;
        comg12 = clam12;
    }
    salp1 = cbet2 * somg12;
    calp1 = comg12 >= 0.0 ? sbet12 + cbet2 * sbet1 * k_sq(somg12) / (1.0 + comg12) : sbet12a - cbet2 * sbet1 * k_sq(somg12) / (1.0 - comg12);
    const double ssig12 = hypot(salp1, calp1), csig12 = sbet1 * sbet2 + cbet1 * cbet2 * comg12;
    if (shortline && ssig12 < K_ETOL2) {
        salp2 = cbet1 * somg12;
        calp2 = sbet12 - cbet1 * sbet2 * (comg12 >= 0.0 ? k_sq(somg12) / (1.0 + comg12) : 1.0 - comg12);
        k_norm2(&salp2, &calp2);
        sig12 = atan2(ssig12, csig12);
    } else if (fabs(K_N) > 0.1 || csig12 >= 0.0 || ssig12 >= 6.0 * fabs(K_N) * K_PI * k_sq(cbet1)) {
        /* nothing to do: zeroth-order spherical approximation is fine */
    } else {
        /* nearly antipodal points: scale lam12 and bet2 to x, y and solve the astroid problem */
        const double lam12x = atan2(-slam12, -clam12);
        const double k2 = k_sq(sbet1) * K_EP2, eps = k2 / (2.0 * (1.0 + sqrt(1.0 + k2)) + k2);
        const double lamscale = K_F * cbet1 * k_A3f(eps) * K_PI, betscale = lamscale * cbet1;
        const double x = lam12x / lamscale, y = sbet12a / betscale;
        if (y > -K_TOL1 && x > -1.0 - K_XTHRESH) {
            salp1 = fmin(1.0, -x);
            calp1 = -sqrt(1.0 - k_sq(salp1));
        } else {
            const double k = k_astroid(x, y);
            const double omg12a = lamscale * (-x * k / (1.0 + k));
            somg12 = sin(omg12a);
            comg12 = -cos(omg12a);
            salp1 = cbet2 * somg12;
            calp1 = sbet12a - cbet2 * sbet1 * k_sq(somg12) / (1.0 - comg12);
        }
    }
    if (!(salp1 <= 0.0)) {
        k_norm2(&salp1, &calp1);
    } else {
        salp1 = 1.0;
        calp1 = 0.0;
    }
    *psalp1 = salp1;
    *pcalp1 = calp1;
    *psalp2 = salp2;
    *pcalp2 = calp2;
    *pdnm = dnm;
    return sig12;
}
/* lambda12(alp1) - lam12 is what Newton's method drives to zero */
KQ double k_lambda12(double sbet1, double cbet1, double dn1, double sbet2, double cbet2, double dn2, double salp1, double calp1, double slam120,
                  double clam120, double* psalp2, double* pcalp2, double* psig12, double* pssig1, double* pcsig1, double* pssig2, double* pcsig2,
                  double* peps, int diffp, double* pdlam12) {
    if (sbet1 == 0.0 && calp1 == 0.0) calp1 = -K_TINY;
    const double salp0 = salp1 * cbet1, calp0 = hypot(calp1, salp1 * sbet1);
    double ssig1 = sbet1, csig1 = calp1 * cbet1;
    const double somg1 = salp0 * sbet1, comg1 = csig1;
    k_norm2(&ssig1, &csig1);
    const double salp2 = cbet2 != cbet1 ? salp0 / cbet2 : salp1;
    const double calp2 = (cbet2 != cbet1 || fabs(sbet2) != -sbet1)
                             ? sqrt(k_sq(calp1 * cbet1) + (cbet1 < -sbet1 ? (cbet2 - cbet1) * (cbet1 + cbet2) : (sbet1 - sbet2) * (sbet1 + sbet2))) / cbet2
                             : fabs(calp1);
    double ssig2 = sbet2, csig2 = calp2 * cbet2;
    const double somg2 = salp0 * sbet2, comg2 = csig2;
    k_norm2(&ssig2, &csig2);
    const double sig12 = atan2(fmax(0.0, csig1 * ssig2 - ssig1 * csig2), csig1 * csig2 + ssig1 * ssig2);
    const double somg12 = fmax(0.0, comg1 * somg2 - somg1 * comg2), comg12 = comg1 * comg2 + somg1 * somg2;
    const double eta = atan2(somg12 * clam120 - comg12 * slam120, comg12 * clam120 + somg12 * slam120);
    const double k2 = k_sq(calp0) * K_EP2, eps = k2 / (2.0 * (1.0 + sqrt(1.0 + k2)) + k2);
    double c3[6];
    k_C3f(eps, c3);
    const double B312 = k_sin_series(ssig2, csig2, c3, 5) - k_sin_series(ssig1, csig1, c3, 5);
    const double domg12 = -K_F * k_A3f(eps) * salp0 * (sig12 + B312);
    const double lam12 = eta + domg12;
    if (diffp) {
        if (calp2 == 0.0) {
            *pdlam12 = -2.0 * K_F1 * dn1 / sbet1;
        } else {
            double s12b, m12b, m0;
            k_lengths(eps, sig12, ssig1, csig1, dn1, ssig2, csig2, dn2, &s12b, &m12b, &m0);
            *pdlam12 = m12b * K_F1 / (calp2 * cbet2);
        }
    }
    *psalp2 = salp2;
    *pcalp2 = calp2;
    *psig12 = sig12;
    *pssig1 = ssig1;
    *pcsig1 = csig1;
    *pssig2 = ssig2;
    *pcsig2 = csig2;
    *peps = eps;
    return lam12;
}
/* metres between (lon1, lat1) and (lon2, lat2), degrees */
KQ double k_geodesic_m(double lon1, double lat1, double lon2, double lat2) {
    double lon12s;
    double lon12 = k_ang_diff(lon1, lon2, &lon12s);
    int lonsign = signbit(lon12) ? -1 : 1;
    lon12 *= lonsign;
    lon12s *= lonsign;
    const double lam12 = lon12 * K_DEGREE;
    double slam12, clam12;
    k_sincosd(lon12, &slam12, &clam12);
    lon12s = (180.0 - lon12) - lon12s;  /* the supplementary longitude difference */
    lat1 = k_ang_round(fabs(lat1) > 90.0 ? NAN : lat1);
    lat2 = k_ang_round(fabs(lat2) > 90.0 ? NAN : lat2);
    const int swapp = fabs(lat1) < fabs(lat2) || lat2 != lat2 ? -1 : 1;
    if (swapp < 0) {
        lonsign *= -1;
        const double t = lat1;
        lat1 = lat2;
        lat2 = t;
    }
    const int latsign = signbit(lat1) ? 1 : -1;
    lat1 *= latsign;
    lat2 *= latsign;
    double sbet1, cbet1, sbet2, cbet2;
    k_sincosd(lat1, &sbet1, &cbet1);
    sbet1 *= K_F1;
    k_norm2(&sbet1, &cbet1);
    cbet1 = fmax(K_TINY, cbet1);
    k_sincosd(lat2, &sbet2, &cbet2);
    sbet2 *= K_F1;
    k_norm2(&sbet2, &cbet2);
    cbet2 = fmax(K_TINY, cbet2);
    if (cbet1 < -sbet1) {
        if (cbet2 == cbet1) sbet2 = copysign(sbet1, sbet2);
    } else {
        if (fabs(sbet2) == -sbet1) cbet2 = cbet1;
    }
    const double dn1 = sqrt(1.0 + K_EP2 * k_sq(sbet1)), dn2 = sqrt(1.0 + K_EP2 * k_sq(sbet2));
    double s12x = 0.0, m12x = 0.0, sig12 = 0.0;
    int meridian = lat1 == -90.0 || slam12 == 0.0;
    if (meridian) {
        const double calp1 = clam12, calp2 = 1.0;
        const double ssig1 = sbet1, csig1 = calp1 * cbet1, ssig2 = sbet2, csig2 = calp2 * cbet2;
        sig12 = atan2(fmax(0.0, csig1 * ssig2 - ssig1 * csig2), csig1 * csig2 + ssig1 * ssig2);
        double m0;
        k_lengths(K_N, sig12, ssig1, csig1, dn1, ssig2, csig2, dn2, &s12x, &m12x, &m0);
        if (sig12 < 1.0 || m12x >= 0.0) {
            if (sig12 < 3.0 * K_TINY || (sig12 < K_TOL0 && (s12x < 0.0 || m12x < 0.0))) sig12 = m12x = s12x = 0.0;
            s12x *= K_B;
        } else {
            meridian = 0;  /* m12 < 0: the prolate case, or the geodesic runs over a pole the long way */
        }
    }
    if (!meridian && sbet1 == 0.0 && lon12s >= K_F * 180.0) {
        s12x = K_A * lam12;  /* along the equator */
    } else if (!meridian) {
        double salp1, calp1, salp2, calp2, dnm;
        sig12 = k_inverse_start(sbet1, cbet1, dn1, sbet2, cbet2, dn2, lam12, slam12, clam12, &salp1, &calp1, &salp2, &calp2, &dnm);
        if (sig12 >= 0.0) {
            s12x = sig12 * K_B * dnm;  /* short line */
        } else {
            double ssig1 = 0.0, csig1 = 0.0, ssig2 = 0.0, csig2 = 0.0, eps = 0.0;
            double salp1a = K_TINY, calp1a = 1.0, salp1b = K_TINY, calp1b = -1.0;
            int tripn = 0, tripb = 0;
            for (int numit = 0; numit < K_MAXIT2; ++numit) {
                double dv = 0.0;
                const double v = k_lambda12(sbet1, cbet1, dn1, sbet2, cbet2, dn2, salp1, calp1, slam12, clam12, &salp2, &calp2, &sig12, &ssig1, &csig1,
                                            &ssig2, &csig2, &eps, numit < K_MAXIT1, &dv);
                if (tripb || !(fabs(v) >= (tripn ? 8.0 : 1.0) * K_TOL0) || numit == K_MAXIT2 - 1) break;
                if (v > 0.0 && (numit > K_MAXIT1 || calp1 / salp1 > calp1b / salp1b)) {
                    salp1b = salp1;
                    calp1b = calp1;
                } else if (v < 0.0 && (numit > K_MAXIT1 || calp1 / salp1 < calp1a / salp1a)) {
                    salp1a = salp1;
                    calp1a = calp1;
                }
                if (numit < K_MAXIT1 && dv > 0.0) {
                    const double dalp1 = -v / dv;
                    if (fabs(dalp1) < K_PI) {
                        const double sdalp1 = sin(dalp1), cdalp1 = cos(dalp1), nsalp1 = salp1 * cdalp1 + calp1 * sdalp1;
                        if (nsalp1 > 0.0) {
                            calp1 = calp1 * cdalp1 - salp1 * sdalp1;
                            salp1 = nsalp1;
                            k_norm2(&salp1, &calp1);
                            tripn = fabs(v) <= 16.0 * K_TOL0;
                            continue;
                        }
                    }
                }
                salp1 = (salp1a + salp1b) / 2.0;
                calp1 = (calp1a + calp1b) / 2.0;
                k_norm2(&salp1, &calp1);
                tripn = 0;
                tripb = (fabs(salp1a - salp1) + (calp1a - calp1) < K_TOLB || fabs(salp1 - salp1b) + (calp1 - calp1b) < K_TOLB);
            }
            double m0;
            k_lengths(eps, sig12, ssig1, csig1, dn1, ssig2, csig2, dn2, &s12x, &m12x, &m0);
            s12x *= K_B;
        }
    }
    return 0.0 + s12x;
}
#undef KQ
#undef K_A
#undef K_F
#undef K_F1
#undef K_E2
#undef K_EP2
#undef K_N
#undef K_B
#undef K_PI
#undef K_DEGREE
#undef K_TINY
#undef K_TOL0
#undef K_TOL1
#undef K_TOL2
#undef K_TOLB
#undef K_XTHRESH
#undef K_ETOL2
#undef K_MAXIT1
#undef K_MAXIT2

}  // namespace karney
}  // namespace gpk
