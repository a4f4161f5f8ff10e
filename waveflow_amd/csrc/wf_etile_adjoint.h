// wf_etile_adjoint.h -- reverse mode of the head algebra of the matrix-core local-energy path (wf_kernels_etile.hip), for the parameter
// gradients of psi and of its Laplacian on the matrix cores (vqmc.py:193-221: value_and_grad(loss_fn_efficient)).
//
// The forward program of one net, per walker (see k_efused): the conditioner leaves Taylor triples (o, o', o'') in s = u_0 of the head's
// pre-activations; the head sums scalars over its rows (separable in s and t = u_1); quotients, logarithms and the change to (x0, x1) jets
// happen once per walker on two-variable Taylor elements.  The functions below restate that once-per-walker part with every intermediate
// kept, and give its PULLBACK: from the adjoints of the outputs to the adjoints of the row sums and of the input jets.  The program is
// differentiated as written (truncated algebra included), which is what reverse mode over the ring arithmetic of the wave sweeps does too:
// no third derivative of the conditioner appears; the table lerps keep the reference's rule (d/dt of the order-k lerp is the order-(k+1)
// lerp, clamped to the last cached order 3: isplines_jax.py:60-66 and JAX's index clamp).
//
// Scalar type S: float in the kernels, double in the CPU test (tests/test_etile_adjoint.py compiles this header with g++ and checks every
// pullback against central differences).
#pragma once

#if defined(__HIPCC__)
#define WF_HD __host__ __device__ __forceinline__
#else
#define WF_HD inline
#endif

namespace wf {
namespace adj {

template <class S> struct Jt {   // value, d/dx0, d/dx1, laplacian / 2
    S v, a, b, h;
};
template <class S> struct T2t {  // f, f_s, f_t, f_ss, f_st, f_tt (true partial derivatives)
    S f, s, t, ss, st, tt;
};
template <class S> WF_HD Jt<S> jzero() { return Jt<S>{S(0), S(0), S(0), S(0)}; }
template <class S> WF_HD T2t<S> t2zero() { return T2t<S>{S(0), S(0), S(0), S(0), S(0), S(0)}; }
template <class S> WF_HD Jt<S> operator+(Jt<S> x, Jt<S> y) { return Jt<S>{x.v + y.v, x.a + y.a, x.b + y.b, x.h + y.h}; }
template <class S> WF_HD Jt<S> operator-(Jt<S> x, Jt<S> y) { return Jt<S>{x.v - y.v, x.a - y.a, x.b - y.b, x.h - y.h}; }
template <class S> WF_HD Jt<S> operator*(Jt<S> x, S c) { return Jt<S>{x.v * c, x.a * c, x.b * c, x.h * c}; }
template <class S> WF_HD Jt<S> jmul(Jt<S> x, Jt<S> y) {
    return Jt<S>{x.v * y.v, x.v * y.a + y.v * x.a, x.v * y.b + y.v * x.b, x.v * y.h + y.v * x.h + (x.a * y.a + x.b * y.b)};
}
// xbar += pullback of jmul with respect to x (the other factor's follows by symmetry)
template <class S> WF_HD void jmul_bwd(Jt<S> y, Jt<S> zb, Jt<S>& xb) {
    xb.v += zb.v * y.v + zb.a * y.a + zb.b * y.b + zb.h * y.h;
    xb.a += zb.a * y.v + zb.h * y.a;
    xb.b += zb.b * y.v + zb.h * y.b;
    xb.h += zb.h * y.v;
}
// f(x) from f, f', f'' at x.v
template <class S> WF_HD Jt<S> japply(Jt<S> x, S f, S f1, S f2) {
    return Jt<S>{f, f1 * x.a, f1 * x.b, f1 * x.h + S(0.5) * f2 * (x.a * x.a + x.b * x.b)};
}
// pullback: adjoints of the three function values (fb[0..2]) and of x's derivative channels; x.v is reached through f, f', f'' by the caller
template <class S> WF_HD void japply_bwd(Jt<S> x, S f1, S f2, Jt<S> yb, S (&fb)[3], Jt<S>& xb) {
    fb[0] += yb.v;
    fb[1] += yb.a * x.a + yb.b * x.b + yb.h * x.h;
    fb[2] += S(0.5) * yb.h * (x.a * x.a + x.b * x.b);
    xb.a += yb.a * f1 + yb.h * f2 * x.a;
    xb.b += yb.b * f1 + yb.h * f2 * x.b;
    xb.h += yb.h * f1;
}
// a scalar function of x.v with derivatives f1, f2, f3 known in closed form: y = japply(x, f, f1, f2); complete pullback
template <class S> WF_HD void jfun_bwd(Jt<S> x, S f1, S f2, S f3, Jt<S> yb, Jt<S>& xb) {
    S fb[3] = {S(0), S(0), S(0)};
    japply_bwd(x, f1, f2, yb, fb, xb);
    xb.v += fb[0] * f1 + fb[1] * f2 + fb[2] * f3;
}

template <class S> WF_HD T2t<S> t2mul(T2t<S> a, T2t<S> b) {
    return T2t<S>{a.f * b.f, a.f * b.s + a.s * b.f, a.f * b.t + a.t * b.f, a.f * b.ss + S(2) * (a.s * b.s) + a.ss * b.f,
                  a.f * b.st + a.s * b.t + a.t * b.s + a.st * b.f, a.f * b.tt + S(2) * (a.t * b.t) + a.tt * b.f};
}
template <class S> WF_HD void t2mul_bwd(T2t<S> b, T2t<S> yb, T2t<S>& ab) {   // with respect to the first factor
    ab.f += yb.f * b.f + yb.s * b.s + yb.t * b.t + yb.ss * b.ss + yb.st * b.st + yb.tt * b.tt;
    ab.s += yb.s * b.f + S(2) * yb.ss * b.s + yb.st * b.t;
    ab.t += yb.t * b.f + yb.st * b.s + S(2) * yb.tt * b.t;
    ab.ss += yb.ss * b.f;
    ab.st += yb.st * b.f;
    ab.tt += yb.tt * b.f;
}
template <class S> WF_HD T2t<S> t2apply(T2t<S> a, S g0, S g1, S g2) {
    return T2t<S>{g0, g1 * a.s, g1 * a.t, g1 * a.ss + g2 * (a.s * a.s), g1 * a.st + g2 * (a.s * a.t), g1 * a.tt + g2 * (a.t * a.t)};
}
// y = g(a) with g', g'', g''' at a.f: complete pullback
template <class S> WF_HD void t2fun_bwd(T2t<S> a, S g1, S g2, S g3, T2t<S> yb, T2t<S>& ab) {
    const S gb0 = yb.f;
    const S gb1 = yb.s * a.s + yb.t * a.t + yb.ss * a.ss + yb.st * a.st + yb.tt * a.tt;
    const S gb2 = yb.ss * a.s * a.s + yb.st * a.s * a.t + yb.tt * a.t * a.t;
    ab.s += yb.s * g1 + S(2) * yb.ss * g2 * a.s + yb.st * g2 * a.t;
    ab.t += yb.t * g1 + yb.st * g2 * a.s + S(2) * yb.tt * g2 * a.t;
    ab.ss += yb.ss * g1;
    ab.st += yb.st * g1;
    ab.tt += yb.tt * g1;
    ab.f += gb0 * g1 + gb1 * g2 + gb2 * g3;
}
// F(s(x), t(x)) as a jet in (x0, x1)
template <class S> WF_HD Jt<S> t2jet(T2t<S> F, Jt<S> s, Jt<S> t) {
    return Jt<S>{F.f, F.s * s.a + F.t * t.a, F.s * s.b + F.t * t.b,
                 F.s * s.h + F.t * t.h + S(0.5) * (F.ss * (s.a * s.a + s.b * s.b) + S(2) * F.st * (s.a * t.a + s.b * t.b) + F.tt * (t.a * t.a + t.b * t.b))};
}
// pullback with respect to F and to the derivative channels of s and t (their value channels enter through F's construction: caller)
template <class S> WF_HD void t2jet_bwd(T2t<S> F, Jt<S> s, Jt<S> t, Jt<S> yb, T2t<S>& Fb, Jt<S>& sb, Jt<S>& tb) {
    const S ss = s.a * s.a + s.b * s.b, st = s.a * t.a + s.b * t.b, tt = t.a * t.a + t.b * t.b;
    Fb.f += yb.v;
    Fb.s += yb.a * s.a + yb.b * s.b + yb.h * s.h;
    Fb.t += yb.a * t.a + yb.b * t.b + yb.h * t.h;
    Fb.ss += S(0.5) * yb.h * ss;
    Fb.st += yb.h * st;
    Fb.tt += S(0.5) * yb.h * tt;
    sb.a += yb.a * F.s + yb.h * (F.ss * s.a + F.st * t.a);
    sb.b += yb.b * F.s + yb.h * (F.ss * s.b + F.st * t.b);
    sb.h += yb.h * F.s;
    tb.a += yb.a * F.t + yb.h * (F.st * s.a + F.tt * t.a);
    tb.b += yb.b * F.t + yb.h * (F.st * s.b + F.tt * t.b);
    tb.h += yb.h * F.t;
}

// r(x) = 1 / (2^x + 1) of a pre-activation triple (x, x', x'') in s: (r, r' x', r' x'' + r'' x'^2); r given (computed by the caller)
template <class S> struct RDeriv {
    S r1, r2, r3;   // r', r'', r''' at x
};
template <class S> WF_HD RDeriv<S> r_derivs(S r) {
    const S L = S(0.6931471805599453);
    const S q = r * (S(1) - r);                    // r' = -L q
    const S r1 = -L * q;
    const S r2 = -L * r1 * (S(1) - S(2) * r);      // (q)' = r'(1 - 2r)
    const S r3 = -L * (r2 * (S(1) - S(2) * r) - S(2) * r1 * r1);
    return RDeriv<S>{r1, r2, r3};
}
template <class S> WF_HD void r_triple(S r, S x1, S x2, S& v0, S& v1, S& v2) {
    const RDeriv<S> d = r_derivs(r);
    v0 = r;
    v1 = d.r1 * x1;
    v2 = d.r1 * x2 + d.r2 * x1 * x1;
}
// pullback of r_triple.  With q = r (1 - r), A = -L (1 - 2 r): r' = -L q, r'' = A r', r''' = r' (A^2 - 2 L^2 q), so every adjoint carries the factor r'
// (15 operations instead of the 23 of the literal form: the activation pullbacks are half of the reverse kernel's vector work)
template <class S> WF_HD void r_triple_bwd(S r, S x1, S x2, S vb0, S vb1, S vb2, S& xb0, S& xb1, S& xb2) {
    const S L = S(0.6931471805599453);
    const S q = r - r * r;
    const S r1 = -L * q;
    const S A = S(2) * L * r - L;
    const S t = x1 * vb2;
    const S u = x1 * vb1 + x2 * vb2;
    const S B = A * A - S(2) * L * L * q;
    xb2 = vb2 * r1;
    xb1 = r1 * (vb1 + S(2) * (A * t));
    xb0 = r1 * (vb0 + A * u + B * (x1 * t));
}

// ------------------------------------------------------------------------------------------------ flow head (made.py:66-81)
// Row sums of one output dimension (see k_efused / flow_rows), extended by the orders the pullback needs:
//   S[a] = sum v_j^(a), Qv[a] = sum g_j v_j^(a) (a = 0..2), R[k] = sum g_j T_j^(k) (k = 0..3),
//   V0[k] = sum v_j g_j T_j^(k) (k = 0..3), V1[k] = sum v_j' g_j T_j^(k) (k = 0..3), V2[k] = sum v_j'' g_j T_j^(k) (k = 0..2)
template <class S> struct FlowSumsT {
    S s[3], qv[3], r[4], v0[4], v1[4], v2[3];
};
template <class S> WF_HD FlowSumsT<S> flow_sums_zero() {
    FlowSumsT<S> z;
    for (int i = 0; i < 3; ++i) { z.s[i] = S(0); z.qv[i] = S(0); z.v2[i] = S(0); }
    for (int i = 0; i < 4; ++i) { z.r[i] = S(0); z.v0[i] = S(0); z.v1[i] = S(0); }
    return z;
}
// y = N_0 / Q as a jet, dl = log(N_1 / Q + 1e-7) as a jet; N_k(s, t) = V_k / S + reg R_k, Q(s) = Qv / S + reg G.
// The argument jets: sj carries s (= u_0, the conditioner's input), tj carries t (the spline's argument).
template <class S> struct FlowHeadFwd {
    T2t<S> iS, Q, rQ, N0, N1, y, d, dl;   // d = N1 * rQ + 1e-7, dl = log d
};
template <class S> WF_HD FlowHeadFwd<S> flow_head_fwd(const FlowSumsT<S>& a, S G, S reg, Jt<S> sj, Jt<S> tj, Jt<S>& y_out, Jt<S>& dl_out) {
    FlowHeadFwd<S> w;
    const T2t<S> St{a.s[0], a.s[1], S(0), a.s[2], S(0), S(0)};
    const S g = S(1) / St.f;
    w.iS = t2apply(St, g, -g * g, S(2) * g * g * g);
    w.Q = t2mul(T2t<S>{a.qv[0], a.qv[1], S(0), a.qv[2], S(0), S(0)}, w.iS);
    w.Q.f += reg * G;
    const S q = S(1) / w.Q.f;
    w.rQ = t2apply(w.Q, q, -q * q, S(2) * q * q * q);
    w.N0 = t2mul(T2t<S>{a.v0[0], a.v1[0], a.v0[1], a.v2[0], a.v1[1], a.v0[2]}, w.iS);
    w.N0.f += reg * a.r[0]; w.N0.t += reg * a.r[1]; w.N0.tt += reg * a.r[2];
    w.N1 = t2mul(T2t<S>{a.v0[1], a.v1[1], a.v0[2], a.v2[1], a.v1[2], a.v0[3]}, w.iS);
    w.N1.f += reg * a.r[1]; w.N1.t += reg * a.r[2]; w.N1.tt += reg * a.r[3];
    w.y = t2mul(w.N0, w.rQ);
    w.d = t2mul(w.N1, w.rQ);
    w.d.f += S(1e-7);
    const S d1 = S(1) / w.d.f;
#if defined(__HIP_DEVICE_COMPILE__)
    w.dl = t2apply(w.d, (S)__logf((float)w.d.f), d1, -d1 * d1);
#else
    w.dl = t2apply(w.d, (S)log((double)w.d.f), d1, -d1 * d1);
#endif
    y_out = t2jet(w.y, sj, tj);
    dl_out = t2jet(w.dl, sj, tj);
    return w;
}
// Pullback: yb, dlb = adjoints of the two output jets.  -> ab (row sums), sb / tb (derivative channels of the argument jets), and
// tvb = adjoint of t's VALUE channel (through the tables: d/dt of a sum over T^(k) is the sum over T^(k+1), clamped at order 3).
template <class S> WF_HD void flow_head_bwd(const FlowSumsT<S>& a, const FlowHeadFwd<S>& w, S G, S reg, Jt<S> sj, Jt<S> tj, Jt<S> yb, Jt<S> dlb,
                                            FlowSumsT<S>& ab, Jt<S>& sb, Jt<S>& tb, S& tvb) {
    (void)G;
    T2t<S> yB = t2zero<S>(), dlB = t2zero<S>();
    t2jet_bwd(w.y, sj, tj, yb, yB, sb, tb);
    t2jet_bwd(w.dl, sj, tj, dlb, dlB, sb, tb);
    // dl = log(d)
    T2t<S> dB = t2zero<S>();
    {
        const S d1 = S(1) / w.d.f;
        t2fun_bwd(w.d, d1, -d1 * d1, S(2) * d1 * d1 * d1, dlB, dB);
    }
    // y = N0 * rQ, d = N1 * rQ (+ const)
    T2t<S> N0B = t2zero<S>(), N1B = t2zero<S>(), rQB = t2zero<S>();
    t2mul_bwd(w.rQ, yB, N0B);
    t2mul_bwd(w.N0, yB, rQB);
    t2mul_bwd(w.rQ, dB, N1B);
    t2mul_bwd(w.N1, dB, rQB);
    // rQ = 1 / Q
    T2t<S> QB = t2zero<S>();
    {
        const S q = S(1) / w.Q.f;
        t2fun_bwd(w.Q, -q * q, S(2) * q * q * q, S(-6) * q * q * q * q, rQB, QB);
    }
    // N_k = V_k-element * iS + reg R;  Q = Qv-element * iS + reg G
    T2t<S> iSB = t2zero<S>(), V0B = t2zero<S>(), V1B = t2zero<S>(), QvB = t2zero<S>();
    const T2t<S> V0e{a.v0[0], a.v1[0], a.v0[1], a.v2[0], a.v1[1], a.v0[2]}, V1e{a.v0[1], a.v1[1], a.v0[2], a.v2[1], a.v1[2], a.v0[3]};
    const T2t<S> Qve{a.qv[0], a.qv[1], S(0), a.qv[2], S(0), S(0)};
    t2mul_bwd(w.iS, N0B, V0B); t2mul_bwd(V0e, N0B, iSB);
    t2mul_bwd(w.iS, N1B, V1B); t2mul_bwd(V1e, N1B, iSB);
    t2mul_bwd(w.iS, QB, QvB);  t2mul_bwd(Qve, QB, iSB);
    ab.r[0] += reg * N0B.f; ab.r[1] += reg * N0B.t + reg * N1B.f; ab.r[2] += reg * N0B.tt + reg * N1B.t; ab.r[3] += reg * N1B.tt;
    // element -> sums: V0e = {v0[0], v1[0], v0[1], v2[0], v1[1], v0[2]}, V1e = {v0[1], v1[1], v0[2], v2[1], v1[2], v0[3]}
    ab.v0[0] += V0B.f; ab.v1[0] += V0B.s; ab.v0[1] += V0B.t; ab.v2[0] += V0B.ss; ab.v1[1] += V0B.st; ab.v0[2] += V0B.tt;
    ab.v0[1] += V1B.f; ab.v1[1] += V1B.s; ab.v0[2] += V1B.t; ab.v2[1] += V1B.ss; ab.v1[2] += V1B.st; ab.v0[3] += V1B.tt;
    ab.qv[0] += QvB.f; ab.qv[1] += QvB.s; ab.qv[2] += QvB.ss;
    // iS = 1 / S-element
    T2t<S> SB = t2zero<S>();
    {
        const T2t<S> St{a.s[0], a.s[1], S(0), a.s[2], S(0), S(0)};
        const S g = S(1) / St.f;
        t2fun_bwd(St, -g * g, S(2) * g * g * g, S(-6) * g * g * g * g, iSB, SB);
    }
    ab.s[0] += SB.f; ab.s[1] += SB.s; ab.s[2] += SB.ss;
    // value channel of t: every sum over T^(k) moves with t like the same sum over T^(k+1) (k + 1 clamped to 3)
    tvb += ab.r[0] * a.r[1] + ab.r[1] * a.r[2] + ab.r[2] * a.r[3] + ab.r[3] * a.r[3];
    tvb += ab.v0[0] * a.v0[1] + ab.v0[1] * a.v0[2] + ab.v0[2] * a.v0[3] + ab.v0[3] * a.v0[3];
    tvb += ab.v1[0] * a.v1[1] + ab.v1[1] * a.v1[2] + ab.v1[2] * a.v1[3];
    tvb += ab.v2[0] * a.v2[1] + ab.v2[1] * a.v2[2];
}
// NOTE for callers: ab must hold ONLY this head's contributions when tvb is formed (pass a zeroed ab).

// ------------------------------------------------------------------------------------------------ prior head (wavefunctions.py:54-71)
// Row sums: D0[k] = sum c_i B_i^(k) (k = 0..3), D1[k] = sum c_i' B_i^(k) (k = 0..2), D2[k] = sum c_i'' B_i^(k) (k = 0..1),
// cc = sum c^2, cc1 = sum c c', c1c1 = sum c'^2, cc2 = sum c c''
template <class S> struct PriorSumsT {
    S d0[4], d1[3], d2[2], cc, cc1, c1c1, cc2;
};
template <class S> WF_HD PriorSumsT<S> prior_sums_zero() {
    PriorSumsT<S> z;
    for (int i = 0; i < 4; ++i) z.d0[i] = S(0);
    for (int i = 0; i < 3; ++i) z.d1[i] = S(0);
    z.d2[0] = z.d2[1] = S(0);
    z.cc = z.cc1 = z.c1c1 = z.cc2 = S(0);
    return z;
}
// val = sgn * (c . B) / |c| as a jet: dot(s, t) * rsqrt(N2(s))
template <class S> struct PriorHeadFwd {
    T2t<S> N2, rn, dot, F;
};
template <class S> WF_HD PriorHeadFwd<S> prior_head_fwd(const PriorSumsT<S>& a, S sgn, Jt<S> sj, Jt<S> tj, Jt<S>& val) {
    PriorHeadFwd<S> w;
    w.N2 = T2t<S>{a.cc, S(2) * a.cc1, S(0), S(2) * (a.c1c1 + a.cc2), S(0), S(0)};
#if defined(__HIP_DEVICE_COMPILE__)
    const S g = (S)rsqrtf((float)w.N2.f);
#else
    const S g = S(1) / (S)sqrt((double)w.N2.f);
#endif
    const S q = S(1) / w.N2.f;
    w.rn = t2apply(w.N2, g, S(-0.5) * g * q, S(0.75) * g * q * q);
    w.dot = T2t<S>{a.d0[0], a.d1[0], a.d0[1], a.d2[0], a.d1[1], a.d0[2]};
    w.F = t2mul(w.dot, w.rn);
    val = t2jet(w.F, sj, tj) * sgn;
    return w;
}
template <class S> WF_HD void prior_head_bwd(const PriorSumsT<S>& a, const PriorHeadFwd<S>& w, S sgn, Jt<S> sj, Jt<S> tj, Jt<S> valb, PriorSumsT<S>& ab,
                                             Jt<S>& sb, Jt<S>& tb, S& tvb) {
    T2t<S> FB = t2zero<S>();
    t2jet_bwd(w.F, sj, tj, valb * sgn, FB, sb, tb);
    T2t<S> dotB = t2zero<S>(), rnB = t2zero<S>(), N2B = t2zero<S>();
    t2mul_bwd(w.rn, FB, dotB);
    t2mul_bwd(w.dot, FB, rnB);
    {
        const S g = w.rn.f, q = S(1) / w.N2.f;
        t2fun_bwd(w.N2, S(-0.5) * g * q, S(0.75) * g * q * q, S(-1.875) * g * q * q * q, rnB, N2B);
    }
    ab.d0[0] += dotB.f; ab.d1[0] += dotB.s; ab.d0[1] += dotB.t; ab.d2[0] += dotB.ss; ab.d1[1] += dotB.st; ab.d0[2] += dotB.tt;
    ab.cc += N2B.f; ab.cc1 += S(2) * N2B.s; ab.c1c1 += S(2) * N2B.ss; ab.cc2 += S(2) * N2B.ss;
    tvb += ab.d0[0] * a.d0[1] + ab.d0[1] * a.d0[2] + ab.d0[2] * a.d0[3];
    tvb += ab.d1[0] * a.d1[1] + ab.d1[1] * a.d1[2];
    tvb += ab.d2[0] * a.d2[1];
}

}  // namespace adj
}  // namespace wf
