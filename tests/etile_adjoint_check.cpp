// CPU check of waveflow_amd/csrc/wf_etile_adjoint.h (compiled with g++ by tests/test_etile_adjoint.py): every pullback against central
// differences of its forward function in double precision.  Prints one line per function: "name max_rel_err"; exit code 1 on a mismatch.
#include <cmath>
#include <cstdio>
#include <functional>
#include <random>
#include <vector>

#include "../waveflow_amd/csrc/wf_etile_adjoint.h"

using namespace wf::adj;
using D = double;
using Vec = std::vector<D>;
static std::mt19937_64 rng(12345);
static D rnd(D lo = -1.0, D hi = 1.0) { return lo + (hi - lo) * std::uniform_real_distribution<D>(0, 1)(rng); }

// checks grad (from the pullback, for the scalar L = sum_i wout[i] * f(x)[i]) against central differences
static int check(const char* name, const Vec& x, const std::function<Vec(const Vec&)>& f, const std::function<Vec(const Vec&, const Vec&)>& pullback, int n_out) {
    Vec w(n_out);
    for (auto& v : w) v = rnd();
    const Vec g = pullback(x, w);
    D worst = 0, scale = 0;
    for (size_t i = 0; i < x.size(); ++i) {
        const D h = 1e-6 * std::max(1.0, std::fabs(x[i]));
        Vec xp = x, xm = x;
        xp[i] += h; xm[i] -= h;
        const Vec fp = f(xp), fm = f(xm);
        D fd = 0;
        for (int k = 0; k < n_out; ++k) fd += w[k] * (fp[k] - fm[k]) / (2 * h);
        worst = std::max(worst, std::fabs(fd - g[i]));
        scale = std::max(scale, std::fabs(fd));
    }
    const D rel = worst / std::max(scale, 1e-12);
    printf("%-28s %.3e\n", name, rel);
    return rel < 2e-6 ? 0 : 1;
}

static Jt<D> J4(const Vec& x, int o) { return Jt<D>{x[o], x[o + 1], x[o + 2], x[o + 3]}; }
static void put(Vec& v, int o, Jt<D> j) { v[o] = j.v; v[o + 1] = j.a; v[o + 2] = j.b; v[o + 3] = j.h; }

int main() {
    int bad = 0;
    // ---- r_triple (r as a function of x0: r = 1 / (2^x0 + 1))
    {
        Vec x = {rnd(-2, 2), rnd(), rnd()};
        auto f = [](const Vec& x) { const D r = 1.0 / (std::exp2(x[0]) + 1.0); D v0, v1, v2; r_triple(r, x[1], x[2], v0, v1, v2); return Vec{v0, v1, v2}; };
        auto pb = [](const Vec& x, const Vec& w) { const D r = 1.0 / (std::exp2(x[0]) + 1.0); D a, b, c; r_triple_bwd(r, x[1], x[2], w[0], w[1], w[2], a, b, c); return Vec{a, b, c}; };
        bad += check("r_triple", x, f, pb, 3);
    }
    // ---- jmul
    {
        Vec x(8);
        for (auto& v : x) v = rnd();
        auto f = [](const Vec& x) { Vec o(4); put(o, 0, jmul(J4(x, 0), J4(x, 4))); return o; };
        auto pb = [](const Vec& x, const Vec& w) { Jt<D> xb = jzero<D>(), yb = jzero<D>(); jmul_bwd(J4(x, 4), J4(w, 0), xb); jmul_bwd(J4(x, 0), J4(w, 0), yb); Vec g(8); put(g, 0, xb); put(g, 4, yb); return g; };
        bad += check("jmul", x, f, pb, 4);
    }
    // ---- jfun (log, rcp, exp_half)
    {
        Vec x = {rnd(0.5, 2), rnd(), rnd(), rnd()};
        auto f = [](const Vec& x) { const D v = x[0]; Vec o(4); put(o, 0, japply(J4(x, 0), std::log(v), 1 / v, -1 / (v * v))); return o; };
        auto pb = [](const Vec& x, const Vec& w) { const D v = x[0]; Jt<D> xb = jzero<D>(); jfun_bwd(J4(x, 0), 1 / v, -1 / (v * v), 2 / (v * v * v), J4(w, 0), xb); Vec g(4); put(g, 0, xb); return g; };
        bad += check("jfun(log)", x, f, pb, 4);
        auto f2 = [](const Vec& x) { const D e = std::exp(0.5 * x[0]); Vec o(4); put(o, 0, japply(J4(x, 0), e, 0.5 * e, 0.25 * e)); return o; };
        auto pb2 = [](const Vec& x, const Vec& w) { const D e = std::exp(0.5 * x[0]); Jt<D> xb = jzero<D>(); jfun_bwd(J4(x, 0), 0.5 * e, 0.25 * e, 0.125 * e, J4(w, 0), xb); Vec g(4); put(g, 0, xb); return g; };
        bad += check("jfun(exp_half)", x, f2, pb2, 4);
    }
    // ---- flow head: inputs = 3 + 3 + 4 + 4 + 4 + 3 sums (21), the derivative channels of s and t (6); outputs: two jets (8).
    // The table value channel is exercised through a model of the sums as functions of t: sum_k(t + h) = sum_k + h sum_{k+1} (clamped), which is
    // what tvb encodes: checked by a directional difference below.
    {
        const D G = 3.7, reg = 0.05;
        Vec x(27);
        for (auto& v : x) v = rnd(-0.5, 0.5);
        x[0] = rnd(2, 4);          // S[0] > 0
        x[3] = rnd(1, 3);          // Qv[0] > 0
        x[10] = rnd(1, 2);         // V0[0]
        x[11] = rnd(1, 2);         // V0[1] (the derivative: positive, inside the log)
        auto unpack = [](const Vec& x, FlowSumsT<D>& a, Jt<D>& sj, Jt<D>& tj) {
            int o = 0;
            for (int i = 0; i < 3; ++i) a.s[i] = x[o++];
            for (int i = 0; i < 3; ++i) a.qv[i] = x[o++];
            for (int i = 0; i < 4; ++i) a.r[i] = x[o++];
            for (int i = 0; i < 4; ++i) a.v0[i] = x[o++];
            for (int i = 0; i < 4; ++i) a.v1[i] = x[o++];
            for (int i = 0; i < 3; ++i) a.v2[i] = x[o++];
            sj = Jt<D>{0.3, x[o], x[o + 1], x[o + 2]};
            tj = Jt<D>{0.6, x[o + 3], x[o + 4], x[o + 5]};
        };
        auto f = [&](const Vec& x) { FlowSumsT<D> a; Jt<D> sj, tj, y, dl; unpack(x, a, sj, tj); flow_head_fwd(a, G, reg, sj, tj, y, dl); Vec o(8); put(o, 0, y); put(o, 4, dl); return o; };
        auto pb = [&](const Vec& x, const Vec& w) {
            FlowSumsT<D> a, ab = flow_sums_zero<D>(); Jt<D> sj, tj, y, dl, sb = jzero<D>(), tb = jzero<D>(); D tvb = 0;
            unpack(x, a, sj, tj);
            const FlowHeadFwd<D> fw = flow_head_fwd(a, G, reg, sj, tj, y, dl);
            flow_head_bwd(a, fw, G, reg, sj, tj, J4(w, 0), J4(w, 4), ab, sb, tb, tvb);
            Vec g; 
            for (int i = 0; i < 3; ++i) g.push_back(ab.s[i]);
            for (int i = 0; i < 3; ++i) g.push_back(ab.qv[i]);
            for (int i = 0; i < 4; ++i) g.push_back(ab.r[i]);
            for (int i = 0; i < 4; ++i) g.push_back(ab.v0[i]);
            for (int i = 0; i < 4; ++i) g.push_back(ab.v1[i]);
            for (int i = 0; i < 3; ++i) g.push_back(ab.v2[i]);
            g.push_back(sb.a); g.push_back(sb.b); g.push_back(sb.h); g.push_back(tb.a); g.push_back(tb.b); g.push_back(tb.h);
            return g;
        };
        // v1[3] and v2[2] are read by the pullback's t-channel only: their own adjoints are zero, and so are the differences
        bad += check("flow_head", x, f, pb, 8);
        // tvb: move every sum along t by its next order
        {
            Vec w(8);
            for (auto& v : w) v = rnd();
            FlowSumsT<D> a, ab = flow_sums_zero<D>(); Jt<D> sj, tj, y, dl, sb = jzero<D>(), tb = jzero<D>(); D tvb = 0;
            unpack(x, a, sj, tj);
            const FlowHeadFwd<D> fw = flow_head_fwd(a, G, reg, sj, tj, y, dl);
            flow_head_bwd(a, fw, G, reg, sj, tj, J4(w, 0), J4(w, 4), ab, sb, tb, tvb);
            auto moved = [&](D h) {
                FlowSumsT<D> m = a;
                for (int k = 0; k < 4; ++k) { m.r[k] += h * a.r[std::min(k + 1, 3)]; m.v0[k] += h * a.v0[std::min(k + 1, 3)]; }
                for (int k = 0; k < 3; ++k) m.v1[k] += h * a.v1[k + 1];
                for (int k = 0; k < 2; ++k) m.v2[k] += h * a.v2[k + 1];
                Jt<D> yy, dd; flow_head_fwd(m, G, reg, sj, tj, yy, dd);
                Vec o(8); put(o, 0, yy); put(o, 4, dd);
                D L = 0; for (int k = 0; k < 8; ++k) L += w[k] * o[k];
                return L;
            };
            const D h = 1e-6, fd = (moved(h) - moved(-h)) / (2 * h);
            const D rel = std::fabs(fd - tvb) / std::max(std::fabs(fd), 1e-12);
            printf("%-28s %.3e\n", "flow_head t-value channel", rel);
            bad += rel < 2e-6 ? 0 : 1;
        }
    }
    // ---- prior head
    {
        Vec x(19);
        for (auto& v : x) v = rnd(-0.5, 0.5);
        x[9] = rnd(1, 2);   // cc > 0
        auto unpack = [](const Vec& x, PriorSumsT<D>& a, Jt<D>& sj, Jt<D>& tj) {
            int o = 0;
            for (int i = 0; i < 4; ++i) a.d0[i] = x[o++];
            for (int i = 0; i < 3; ++i) a.d1[i] = x[o++];
            for (int i = 0; i < 2; ++i) a.d2[i] = x[o++];
            a.cc = x[o++]; a.cc1 = x[o++]; a.c1c1 = x[o++]; a.cc2 = x[o++];
            sj = Jt<D>{0.3, x[o], x[o + 1], x[o + 2]};
            tj = Jt<D>{0.6, x[o + 3], x[o + 4], x[o + 5]};
        };
        const D sgn = -1.0;
        auto f = [&](const Vec& x) { PriorSumsT<D> a; Jt<D> sj, tj, v; unpack(x, a, sj, tj); prior_head_fwd(a, sgn, sj, tj, v); Vec o(4); put(o, 0, v); return o; };
        auto pb = [&](const Vec& x, const Vec& w) {
            PriorSumsT<D> a, ab = prior_sums_zero<D>(); Jt<D> sj, tj, v, sb = jzero<D>(), tb = jzero<D>(); D tvb = 0;
            unpack(x, a, sj, tj);
            const PriorHeadFwd<D> fw = prior_head_fwd(a, sgn, sj, tj, v);
            prior_head_bwd(a, fw, sgn, sj, tj, J4(w, 0), ab, sb, tb, tvb);
            Vec g;
            for (int i = 0; i < 4; ++i) g.push_back(ab.d0[i]);
            for (int i = 0; i < 3; ++i) g.push_back(ab.d1[i]);
            for (int i = 0; i < 2; ++i) g.push_back(ab.d2[i]);
            g.push_back(ab.cc); g.push_back(ab.cc1); g.push_back(ab.c1c1); g.push_back(ab.cc2);
            g.push_back(sb.a); g.push_back(sb.b); g.push_back(sb.h); g.push_back(tb.a); g.push_back(tb.b); g.push_back(tb.h);
            return g;
        };
        bad += check("prior_head", x, f, pb, 4);
    }
    if (bad) printf("FAILED: %d\n", bad);
    return bad ? 1 : 0;
}
