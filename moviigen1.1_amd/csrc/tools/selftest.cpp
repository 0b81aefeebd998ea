// Standalone GPU self-test + micro-bench for libmoviigen_hip.so (no torch, no oracle import).
// TEST INFRASTRUCTURE: each kernel is compared with a straightforward host loop written here
// (double accumulation).  Usage:  mg_selftest [quick|full]
//   quick: correctness at small shapes;  full: + sampled correctness and timing at 14B / 720p shapes
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "moviigen_hip.h"
extern "C" void mg_gemm_debug_profile(unsigned long long* dev_buf);
extern "C" void mg_attn_w64_debug(int flags);
extern "C" void mg_attn_w64_profile(unsigned long long* dev_buf);
extern "C" void mg_gemm5_debug_profile(unsigned long long* dev_buf);
extern "C" void mg_attn_w64_flag_counter(unsigned* dev_counter);

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(2);                                                               \
        }                                                                          \
    } while (0)

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static float frand() {  // uniform [-1,1)
    rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull;
    return (float)((rng_state >> 40) & 0xFFFFFF) / 8388608.0f - 1.0f;
}
static uint16_t f2bf(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static float bf2f(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static float rbf(float f) { return bf2f(f2bf(f)); }

template <typename T>
struct Dev {
    T* p = nullptr;
    size_t n = 0;
    Dev() {}
    explicit Dev(size_t n_) : n(n_) { CK(hipMalloc(&p, n * sizeof(T))); }
    Dev(const std::vector<T>& h) : n(h.size()) {
        CK(hipMalloc(&p, n * sizeof(T)));
        CK(hipMemcpy(p, h.data(), n * sizeof(T), hipMemcpyHostToDevice));
    }
    ~Dev() { if (p) (void)hipFree(p); }
    std::vector<T> host() const {
        std::vector<T> h(n);
        CK(hipMemcpy(h.data(), p, n * sizeof(T), hipMemcpyDeviceToHost));
        return h;
    }
    void zero() { CK(hipMemset(p, 0, n * sizeof(T))); }
    Dev(const Dev&) = delete;
    Dev& operator=(const Dev&) = delete;
};

// the caller-owned ticket workspace of mg_attn_fwd_bf16_hd128* (include/moviigen_hip.h): this client allocates and zeroes it ONCE, outside
// any launch; every launch of the (single) stream shares it
static void* attn_ws() {
    static void* ws = nullptr;
    if (!ws) {
        CK(hipMalloc(&ws, (size_t)mg_attn_workspace_bytes()));
        CK(hipMemset(ws, 0, (size_t)mg_attn_workspace_bytes()));
        CK(hipDeviceSynchronize());
    }
    return ws;
}

static int n_fail = 0;
static void report(const char* name, double err, double tol) {
    const bool ok = (err <= tol) && !isnan(err);
    printf("[%s] %-44s max_err=%.3e tol=%.1e\n", ok ? "PASS" : "FAIL", name, err, tol);
    if (!ok) ++n_fail;
    fflush(stdout);
}
static std::vector<float> randf(size_t n, float s = 1.f) {
    std::vector<float> v(n);
    for (auto& x : v) x = frand() * s;
    return v;
}
static std::vector<uint16_t> randbf(size_t n, float s = 1.f) {
    std::vector<uint16_t> v(n);
    for (auto& x : v) x = f2bf(frand() * s);
    return v;
}
static double gelu_tanh_ref(double x) {
    return 0.5 * x * (1.0 + tanh(0.7978845608028654 * (x + 0.044715 * x * x * x)));
}

template <typename F>
static float time_ms(F f, int iters) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    CK(hipEventDestroy(a));
    CK(hipEventDestroy(b));
    return ms / iters;
}

// ------------------------------------------------------------------------------------------------
static void test_ln(int rows, int dim, int add_one, int round_bf, int out_f32) {
    auto x = randf((size_t)rows * dim, 3.f), sc = randf(dim), sh = randf(dim);
    Dev<float> dx(x), dsc(sc), dsh(sh);
    Dev<float> of((size_t)rows * dim);
    Dev<uint16_t> ob((size_t)rows * dim);
    int rc = mg_ln_modulate(dx.p, dim, rows, dim, dsc.p, dsh.p, add_one, 1e-6f, round_bf,
                            out_f32 ? (void*)of.p : (void*)ob.p, out_f32, dim, 0);
    CK(hipDeviceSynchronize());
    double err = rc ? 1e9 : 0;
    auto hf = of.host();
    auto hb = ob.host();
    for (int r = 0; r < rows; ++r) {
        double mean = 0, var = 0;
        for (int c = 0; c < dim; ++c) mean += x[(size_t)r * dim + c];
        mean /= dim;
        for (int c = 0; c < dim; ++c) { double d = x[(size_t)r * dim + c] - mean; var += d * d; }
        var /= dim;
        double rstd = 1.0 / sqrt(var + 1e-6);
        for (int c = 0; c < dim; ++c) {
            double y = (x[(size_t)r * dim + c] - mean) * rstd;
            if (round_bf) y = rbf((float)y);
            double ref = y * (add_one ? 1.0 + sc[c] : sc[c]) + sh[c];
            double got = out_f32 ? hf[(size_t)r * dim + c] : bf2f(hb[(size_t)r * dim + c]);
            double tol_scale = out_f32 ? 1.0 : 1.0;
            err = fmax(err, fabs(got - ref) / tol_scale);
        }
    }
    char nm[128];
    snprintf(nm, sizeof nm, "ln_modulate r%d d%d add1=%d rnd=%d f32=%d", rows, dim, add_one, round_bf, out_f32);
    report(nm, err, out_f32 ? (round_bf ? 3e-2 : 2e-5) : 4e-2);
}

static void test_rmsnorm_rope(int rows, int dim, int hd, int F, int H, int W, int64_t pos0, bool rope) {
    auto x = randbf((size_t)rows * dim, 2.f);
    auto w = randf(dim);
    const int c = hd / 2, c1 = c / 3, c0 = c - 2 * c1;
    std::vector<float> cs((size_t)2 * (F * c0 + H * c1 + W * c1));
    {
        size_t o = 0;
        auto fill = [&](int n, int cnt, int dimax) {
            for (int p = 0; p < n; ++p)
                for (int j = 0; j < cnt; ++j) {
                    double fr = 1.0 / pow(10000.0, (double)(2 * j) / dimax);
                    cs[o++] = (float)cos(p * fr);
                    cs[o++] = (float)sin(p * fr);
                }
        };
        fill(F, c0, 2 * c0);
        fill(H, c1, 2 * c1);
        fill(W, c1, 2 * c1);
    }
    Dev<uint16_t> dx(x), dout((size_t)rows * dim);
    Dev<float> dw(w), dcs(cs);
    int rc = mg_rmsnorm_rope_bf16(dx.p, dim, dout.p, dim, rows, dim, dw.p, 1e-6f, hd, rope ? dcs.p : nullptr,
                                  F, H, W, pos0, 1.0f, 0);
    CK(hipDeviceSynchronize());
    auto got = dout.host();
    double err = rc ? 1e9 : 0;
    for (int r = 0; r < rows; ++r) {
        double ss = 0;
        for (int j = 0; j < dim; ++j) { double v = bf2f(x[(size_t)r * dim + j]); ss += v * v; }
        double rr = 1.0 / sqrt(ss / dim + 1e-6);
        int64_t tok = pos0 + r;
        bool dr = rope && tok < (int64_t)F * H * W;
        int pf = 0, ph = 0, pw = 0;
        if (dr) { pf = tok / (H * W); ph = (tok % (H * W)) / W; pw = tok % W; }
        for (int j = 0; j < dim; j += 2) {
            double a = (double)rbf((float)(bf2f(x[(size_t)r * dim + j]) * rr)) * w[j];
            double b = (double)rbf((float)(bf2f(x[(size_t)r * dim + j + 1]) * rr)) * w[j + 1];
            double oa = a, ob = b;
            if (dr) {
                int p = (j % hd) / 2;
                double co, si;
                if (p < c0) { co = cs[2 * ((size_t)pf * c0 + p)]; si = cs[2 * ((size_t)pf * c0 + p) + 1]; }
                else if (p < c0 + c1) { size_t o = (size_t)F * c0 + (size_t)ph * c1 + (p - c0); co = cs[2 * o]; si = cs[2 * o + 1]; }
                else { size_t o = (size_t)F * c0 + (size_t)H * c1 + (size_t)pw * c1 + (p - c0 - c1); co = cs[2 * o]; si = cs[2 * o + 1]; }
                oa = a * co - b * si;
                ob = a * si + b * co;
            }
            err = fmax(err, fabs(bf2f(got[(size_t)r * dim + j]) - oa));
            err = fmax(err, fabs(bf2f(got[(size_t)r * dim + j + 1]) - ob));
        }
    }
    char nm[128];
    snprintf(nm, sizeof nm, "rmsnorm_rope r%d d%d hd%d rope=%d pos0=%lld", rows, dim, hd, (int)rope, (long long)pos0);
    report(nm, err, 6e-2);
}

static void test_pack(int64_t L, int heads) {
    const int64_t nt = (L + 63) / 64;
    const int ldv = heads * 128 * 3;  // strided like a fused QKV buffer
    auto kv = randbf((size_t)L * ldv);
    Dev<uint16_t> dkv(kv), dkp((size_t)heads * nt * 8192), dvp((size_t)heads * nt * 8192);
    CK(hipMemset(dkp.p, 0xff, dkp.n * 2));
    CK(hipMemset(dvp.p, 0xff, dvp.n * 2));
    int rc = mg_pack_kv_bf16(dkv.p + heads * 128, ldv, dkv.p + 2 * heads * 128, ldv, L, heads, 128, dkp.p, dvp.p, 0);
    CK(hipDeviceSynchronize());
    auto gk = dkp.host(), gv = dvp.host();
    double err = rc ? 1e9 : 0;
    for (int h = 0; h < heads; ++h)
        for (int64_t key = 0; key < nt * 64; ++key)
            for (int d = 0; d < 128; ++d) {
                const int64_t t = key / 64, r = key % 64;
                const float rk = key < L ? bf2f(kv[(size_t)key * ldv + heads * 128 + h * 128 + d]) : 0.f;
                const float rv = key < L ? bf2f(kv[(size_t)key * ldv + 2 * heads * 128 + h * 128 + d]) : 0.f;
                const size_t tb = ((size_t)h * nt + t) * 8192;
                const int64_t row = (r & 32) | ((r & 4) << 2) | ((r & 24) >> 1) | (r & 3);   // m16 K row order
                err = fmax(err, fabs(bf2f(gk[tb + (d / 8) * 512 + row * 8 + (d % 8)]) - rk));
                err = fmax(err, fabs(bf2f(gv[tb + (r / 8) * 1024 + d * 8 + (r % 8)]) - rv));
            }
    char nm[128];
    snprintf(nm, sizeof nm, "pack_kv L%lld heads%d", (long long)L, heads);
    report(nm, err, 0.0);
}

// sampled GEMM check; nsamp<=0 -> full check
static void test_gemm(int64_t M, int N, int K, int epi, int nsamp, bool timeit) {
    auto A = randbf((size_t)M * K), Wt = randbf((size_t)N * K, 0.05f);
    auto bias = randf(N), gate = randf(N);
    const int64_t ldo = (N + 3) / 4 * 4;
    std::vector<float> resid0;
    Dev<uint16_t> dA(A), dW(Wt), ob((size_t)M * ldo);
    Dev<float> db(bias), dg(gate), of((size_t)M * ldo);
    if (epi == MG_EPI_GATE_RESID_F32) {
        resid0 = randf((size_t)M * ldo);
        CK(hipMemcpy(of.p, resid0.data(), resid0.size() * 4, hipMemcpyHostToDevice));
    }
    const bool f32out = epi >= 2;
    void* outp = f32out ? (void*)of.p : (void*)ob.p;
    int rc = mg_gemm_bf16(dA.p, K, dW.p, K, db.p, M, N, K, epi, outp, ldo, dg.p, 0);
    CK(hipDeviceSynchronize());
    std::vector<float> hf;
    std::vector<uint16_t> hb;
    if (f32out) hf = of.host(); else hb = ob.host();
    double err = rc ? 1e9 : 0;
    const int64_t total = M * N;
    const int64_t cnt = nsamp > 0 ? nsamp : total;
    int n_bad = 0;
    for (int64_t s = 0; s < cnt; ++s) {
        int64_t m, n;
        if (nsamp > 0) {
            rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull;
            m = (int64_t)((rng_state >> 20) % (uint64_t)M);
            n = (int)((rng_state >> 44) % (uint64_t)N);
            if (s < 64) { m = M - 1 - (s & 7); n = N - 1 - (s >> 3); }  // exercise the edges
        } else { m = s / N; n = s % N; }
        double acc = 0;
        for (int kx = 0; kx < K; ++kx) acc += (double)bf2f(A[(size_t)m * K + kx]) * bf2f(Wt[(size_t)n * K + kx]);
        double y = rbf((float)(acc + bias[n]));
        double ref, got;
        if (epi == MG_EPI_BIAS_BF16) { ref = y; got = bf2f(hb[(size_t)m * ldo + n]); }
        else if (epi == MG_EPI_BIAS_GELU_BF16) { ref = gelu_tanh_ref(y); got = bf2f(hb[(size_t)m * ldo + n]); }
        else if (epi == MG_EPI_GATE_RESID_F32) { ref = resid0[(size_t)m * ldo + n] + y * gate[n]; got = hf[(size_t)m * ldo + n]; }
        else { ref = y; got = hf[(size_t)m * ldo + n]; }
        err = fmax(err, fabs(got - ref) / fmax(1.0, fabs(ref)));
        if (getenv("MG_GEMM_SHOW_BAD") && n_bad < 12 && fabs(got - ref) / fmax(1.0, fabs(ref)) > 2e-2) {
            ++n_bad;
            printf("    bad (m %lld, n %lld): got %.6f want %.6f  y %.6f%s\n", (long long)m, (long long)n, got, ref, y,
                   epi == MG_EPI_GATE_RESID_F32 ? "" : "");
            if (epi == MG_EPI_GATE_RESID_F32) printf("        resid %.6f gate %.6f\n", resid0[(size_t)m * ldo + n], gate[n]);
        }
    }
    char nm[160];
    snprintf(nm, sizeof nm, "gemm_bf16 M%lld N%d K%d epi%d", (long long)M, N, K, epi);
    report(nm, err, 2e-2);
    if (timeit) {
        float ms = time_ms([&] { mg_gemm_bf16(dA.p, K, dW.p, K, db.p, M, N, K, epi, outp, ldo, dg.p, 0); }, 5);
        printf("    time %.3f ms  -> %.1f TFLOP/s\n", ms, 2.0 * M * N * K / (ms * 1e-3) / 1e12);
    }
}

// A/B of GEMM schedules on ONE set of operands, interleaved rounds (cdna guide 5.4 rule 24): the operands are generated
// once per shape, every (round, variant) is timed over 5 launches; results of all variants are compared with the first
// one's (bit-identical by construction: the schedule never changes the accumulation order).
static void gemm_ab(int64_t M, int N, int K, int epi, const std::vector<int>& variants, int rounds) {
    auto A = randbf((size_t)M * K), Wt = randbf((size_t)N * K, 0.05f);
    auto bias = randf(N), gate = randf(N);
    const int64_t ldo = (N + 3) / 4 * 4;
    Dev<uint16_t> dA(A), dW(Wt), ob((size_t)M * ldo);
    Dev<float> db(bias), dg(gate), of((size_t)M * ldo), of0((size_t)M * ldo);
    const bool f32out = epi >= 2;
    if (epi == MG_EPI_GATE_RESID_F32) {     // a bounded residual stream: x accumulates over the timed launches
        auto r = randf((size_t)M * ldo);
        CK(hipMemcpy(of0.p, r.data(), r.size() * 4, hipMemcpyHostToDevice));
    }
    void* outp = f32out ? (void*)of.p : (void*)ob.p;
    printf("gemm_ab M=%lld N=%d K=%d epi=%d\n", (long long)M, N, K, epi);
    std::vector<float> first_f;
    std::vector<uint16_t> first_b;
    for (int r = 0; r < rounds; ++r)
        for (int var : variants) {
            mg_gemm_set_variant(var);
            if (f32out) CK(hipMemcpy(of.p, of0.p, of.n * 4, hipMemcpyDeviceToDevice));
            int rc = mg_gemm_bf16(dA.p, K, dW.p, K, db.p, M, N, K, epi, outp, ldo, dg.p, 0);
            CK(hipDeviceSynchronize());
            bool same = rc == 0;
            if (f32out) {
                auto h = of.host();
                if (first_f.empty()) first_f = h; else same = same && !memcmp(h.data(), first_f.data(), h.size() * 4);
            } else {
                auto h = ob.host();
                if (first_b.empty()) first_b = h; else same = same && !memcmp(h.data(), first_b.data(), h.size() * 2);
            }
            if (!same) ++n_fail;
            float ms = time_ms([&] { mg_gemm_bf16(dA.p, K, dW.p, K, db.p, M, N, K, epi, outp, ldo, dg.p, 0); }, 5);
            printf("  [%s] variant %d round %d: %.3f ms  %.1f TFLOP/s\n", same ? "SAME" : "DIFF", var, r, ms, 2.0 * M * N * K / (ms * 1e-3) / 1e12);
            fflush(stdout);
        }
    mg_gemm_set_variant(0);
}

static void test_attn(int64_t Lq, int64_t Lk, int heads, int nsamp, bool timeit, int prescaled = 0) {
    const int64_t ld = (int64_t)heads * 128;
    const int64_t npk = (int64_t)heads * ((Lk + 63) / 64) * 8192;
    auto q = randbf((size_t)Lq * ld, 2.0f), k = randbf((size_t)Lk * ld, 2.0f), v = randbf((size_t)Lk * ld);
    // make the softmax peaky for some rows: a few keys aligned with queries
    for (int i = 0; i < 8 && i < Lk && i < Lq; ++i)
        for (int d = 0; d < 128; ++d) k[(size_t)((i * 37) % Lk) * ld + d] = f2bf(3.f * bf2f(q[(size_t)i * ld + d]));
    Dev<uint16_t> dq(q), dk(k), dv(v), dkp((size_t)npk), dvp((size_t)npk), dout((size_t)Lq * ld);
    CK(hipMemset(dout.p, 0xff, dout.n * 2));
    int rc = mg_pack_kv_bf16(dk.p, ld, dv.p, ld, Lk, heads, 128, dkp.p, dvp.p, 0);
    const float scale = 1.f / sqrtf(128.f);
    // prescaled: the kernel gets bf16(q * scale*log2e) through the pre-scaled entry; the reference below keeps using
    // the unscaled q (this q' is a second rounding: the DiT applies the factor before q's first one)
    std::vector<uint16_t> qs;
    Dev<uint16_t> dqs;
    if (prescaled) {
        qs.resize(q.size());
        for (size_t i = 0; i < q.size(); ++i) qs[i] = f2bf(bf2f(q[i]) * scale * 1.4426950408889634f);
        new (&dqs) Dev<uint16_t>(qs);
    }
    auto run = [&]() {
        return prescaled ? mg_attn_fwd_bf16_hd128_prescaled(dqs.p, ld, dkp.p, dvp.p, dout.p, ld, nullptr, Lq, Lk, heads, 0, attn_ws(), 0)
                         : mg_attn_fwd_bf16_hd128(dq.p, ld, dkp.p, dvp.p, dout.p, ld, Lq, Lk, heads, scale, attn_ws(), 0);
    };
    rc |= run();
    CK(hipDeviceSynchronize());
    auto got = dout.host();
    double err = rc ? 1e9 : 0;
    const int64_t total = Lq * heads;
    const int64_t cnt = nsamp > 0 ? nsamp : total;
    std::vector<double> s(Lk);
    int64_t worst_q = -1; int worst_h = -1, worst_d = -1; double worst_got = 0, worst_ref = 0;
    for (int64_t it = 0; it < cnt; ++it) {
        int64_t qi; int h;
        if (nsamp > 0) {
            rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull;
            qi = (int64_t)((rng_state >> 20) % (uint64_t)Lq);
            h = (int)((rng_state >> 50) % (uint64_t)heads);
            if (it < 8) qi = it;                 // the peaky rows
            else if (it < 16) qi = Lq - 1 - (it - 8);  // tail of the last query block
        } else { qi = it / heads; h = it % heads; }
        double mx = -1e300;
        for (int64_t j = 0; j < Lk; ++j) {
            double a = 0;
            for (int d = 0; d < 128; ++d) a += (double)bf2f(q[(size_t)qi * ld + h * 128 + d]) * bf2f(k[(size_t)j * ld + h * 128 + d]);
            s[j] = a * scale;
            mx = fmax(mx, s[j]);
        }
        double den = 0;
        for (int64_t j = 0; j < Lk; ++j) { s[j] = exp(s[j] - mx); den += s[j]; }
        for (int d = 0; d < 128; ++d) {
            double a = 0;
            for (int64_t j = 0; j < Lk; ++j) a += s[j] * bf2f(v[(size_t)j * ld + h * 128 + d]);
            a /= den;
            const double e1 = fabs(bf2f(got[(size_t)qi * ld + h * 128 + d]) - a);
            if (e1 > err) { err = e1; worst_q = qi; worst_h = h; worst_d = d; worst_got = bf2f(got[(size_t)qi * ld + h * 128 + d]); worst_ref = a; }
        }
    }
    if (err > 2e-2) printf("    worst: query %lld head %d d %d got %g ref %g\n", (long long)worst_q, worst_h, worst_d, worst_got, worst_ref);
    char nm[160];
    snprintf(nm, sizeof nm, "attn_fwd Lq%lld Lk%lld heads%d%s", (long long)Lq, (long long)Lk, heads, prescaled ? " prescaled q" : "");
    report(nm, err, 2e-2);
    if (timeit) {
        float ms = time_ms([&] { run(); }, 3);
        printf("    time %.3f ms  -> %.1f TFLOP/s\n", ms, 4.0 * Lq * Lk * 128 * heads / (ms * 1e-3) / 1e12);
        float mt = time_ms([&] { mg_pack_kv_bf16(dk.p, ld, dv.p, ld, Lk, heads, 128, dkp.p, dvp.p, 0); }, 3);
        printf("    pack_kv %.3f ms\n", mt);
    }
}

// A/B of the attention kernel variants on ONE set of operands, interleaved rounds (cdna guide 5.4 rule 24).
// data 0: uniform [-2,2) q/k (logit sigma 1.3 natural units: every row near-uniform);
// data 1..4: logit sigma 8 / 16 / 24 / 40 natural units (the regime of trained checkpoints and far beyond), an
//   attention-sink key (key 0: +2 sigma on every row) and a per-head gain spread (head h: 0.6 .. 1.0 of the nominal sigma);
// data 5: sigma 24 with the sink at key L/2 (+3 sigma) instead of key 0 — the row maximum is NOT in the first key tile.
// Reports TFLOP/s per variant and round, the count of query blocks that were repeated with swept maxima / sent to the
// exact loop, and sampled-row errors vs a double reference.
static const char* attn_ab_data_name(int data) {
    static const char* names[] = {"uniform", "sigma8+sink+gains", "sigma16+sink+gains", "sigma24+sink+gains", "sigma40+sink+gains",
                                  "sigma24+late-sink+gains"};
    return data >= 0 && data < 6 ? names[data] : "?";
}
static void attn_ab(int64_t L, int heads, int data, const std::vector<int>& variants, int rounds) {
    const int64_t ld = (int64_t)heads * 128;
    const int64_t npk = (int64_t)heads * ((L + 63) / 64) * 8192;
    static const float sigmas[] = {0.f, 8.f, 16.f, 24.f, 40.f, 24.f};
    const float sigma = data >= 1 && data <= 5 ? sigmas[data] : 0.f;
    const float amp = data ? sqrtf(3.f * sigma) : 2.0f;       // uniform[-a,a) has variance a^2/3: logit sigma = a^2/3
    auto q = randbf((size_t)L * ld, amp), k = randbf((size_t)L * ld, amp), v = randbf((size_t)L * ld);
    if (data) {
        const int64_t sink = data == 5 ? L / 2 : 0;
        const float sink_sigmas = data == 5 ? 3.f : 2.f;
        for (int h = 0; h < heads; ++h) {
            const float gain = heads > 1 ? 0.6f + 0.4f * h / (heads - 1) : 1.f;
            for (int64_t i = 0; i < L; ++i) {
                for (int d = 1; d < 128; ++d) q[(size_t)i * ld + h * 128 + d] = f2bf(gain * bf2f(q[(size_t)i * ld + h * 128 + d]));
                q[(size_t)i * ld + h * 128] = f2bf(3.f);
            }
            k[(size_t)sink * ld + h * 128] = f2bf(sink_sigmas * sigma * gain * sqrtf(128.f) / 3.f);    // 3 * this / sqrt(128) = + n sigma
        }
    }
    Dev<uint16_t> dq(q), dk(k), dv(v), dkp((size_t)npk), dvp((size_t)npk), dout((size_t)L * ld);
    Dev<unsigned> cnt(2);
    const int reserve = getenv("MG_ATTN_RESERVE_CUS") ? atoi(getenv("MG_ATTN_RESERVE_CUS")) : 0;   // CUs left free by the pre-scaled entry
    const float scale = 1.f / sqrtf(128.f);
    std::vector<uint16_t> qs(q.size());          // variants >= 10: the pre-scaled entry on bf16(q * scale*log2e)
    for (size_t i = 0; i < q.size(); ++i) qs[i] = f2bf(bf2f(q[i]) * scale * 1.4426950408889634f);
    Dev<uint16_t> dqs(qs);
    int rc = mg_pack_kv_bf16(dk.p, ld, dv.p, ld, L, heads, 128, dkp.p, dvp.p, 0);
    if (rc) { printf("pack_kv failed %d\n", rc); ++n_fail; return; }
    mg_attn_w64_flag_counter(cnt.p);
    const double flop = 4.0 * L * L * 128 * heads;
    printf("attn_ab L=%lld heads=%d data=%d (%s)\n", (long long)L, heads, data, attn_ab_data_name(data));
    // reference rows (double), sampled once — one set per entry point: the general entry attends with q and `scale`, the
    // pre-scaled entry with the operand it is GIVEN, bf16(q * scale * log2 e) (here a second rounding of q; in the DiT the
    // factor enters before q's only rounding), whose scores are base-2 exponents.  Each kernel is held to its own operands.
    const int nsamp = 12;
    std::vector<int64_t> sq(nsamp); std::vector<int> sh(nsamp);
    std::vector<std::vector<double>> ref[2] = {std::vector<std::vector<double>>(nsamp, std::vector<double>(128)),
                                               std::vector<std::vector<double>>(nsamp, std::vector<double>(128))};
    std::vector<double> sc(L);
    double ref_max[2] = {0, 0};
    for (int it = 0; it < nsamp; ++it) {
        rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull;
        sq[it] = it < 2 ? it : (it < 4 ? L - 1 - it : (int64_t)((rng_state >> 20) % (uint64_t)L));
        sh[it] = (int)((rng_state >> 50) % (uint64_t)heads);
        for (int pre = 0; pre < 2; ++pre) {
            const std::vector<uint16_t>& qq = pre ? qs : q;
            const double mul = pre ? 0.6931471805599453 : (double)scale;
            double mx = -1e300;
            for (int64_t j = 0; j < L; ++j) {
                double a = 0;
                for (int d = 0; d < 128; ++d) a += (double)bf2f(qq[(size_t)sq[it] * ld + sh[it] * 128 + d]) * bf2f(k[(size_t)j * ld + sh[it] * 128 + d]);
                sc[j] = a * mul;
                mx = fmax(mx, sc[j]);
            }
            double den = 0;
            for (int64_t j = 0; j < L; ++j) { sc[j] = exp(sc[j] - mx); den += sc[j]; }
            for (int d = 0; d < 128; ++d) {
                double a = 0;
                for (int64_t j = 0; j < L; ++j) a += sc[j] * bf2f(v[(size_t)j * ld + sh[it] * 128 + d]);
                ref[pre][it][d] = a / den;
                ref_max[pre] = fmax(ref_max[pre], fabs(ref[pre][it][d]));
            }
        }
    }
    for (int r = 0; r < rounds; ++r)
        for (int var : variants) {
            // variant: 0 = m16 general entry, 3 = w64; 10 + k = m16 through the pre-scaled entry with debug flags 2*k
            // (k = 1: the alternative filler schedule)
            mg_attn_set_variant(var >= 10 ? 0 : var);
            mg_attn_w64_debug(var >= 10 ? 2 * (var - 10) : 0);
            rc = mg_pack_kv_bf16(dk.p, ld, dv.p, ld, L, heads, 128, dkp.p, dvp.p, 0);   // the K row order follows the kernel
            cnt.zero();
            CK(hipMemset(dout.p, 0xff, dout.n * 2));
            auto run = [&]() {
                return var >= 10 ? mg_attn_fwd_bf16_hd128_prescaled(dqs.p, ld, dkp.p, dvp.p, dout.p, ld, nullptr, L, L, heads, reserve, attn_ws(), 0)
                                 : mg_attn_fwd_bf16_hd128(dq.p, ld, dkp.p, dvp.p, dout.p, ld, L, L, heads, scale, attn_ws(), 0);
            };
            rc |= run();
            CK(hipDeviceSynchronize());
            const unsigned flagged = cnt.host()[0], to_exact = cnt.host()[1];
            float ms = time_ms([&] { run(); }, 3);
            auto got = dout.host();
            double err = rc ? 1e9 : 0;
            const int pre = var >= 10;
            for (int it = 0; it < nsamp; ++it)
                for (int d = 0; d < 128; ++d)
                    err = fmax(err, fabs(bf2f(got[(size_t)sq[it] * ld + sh[it] * 128 + d]) - ref[pre][it][d]));
            const bool ok = err <= 2e-2 * fmax(ref_max[pre], 1e-3);
            if (!ok) ++n_fail;
            printf("  [%s] variant %d round %d: %.3f ms  %.1f TFLOP/s  flagged blocks %u (exact loop %u) of %lld  max_err %.3e (ref max %.3e)\n",
                   ok ? "PASS" : "FAIL", var, r, ms, flop / (ms * 1e-3) / 1e12, flagged, to_exact, (long long)((L + 255) / 256) * heads, err, ref_max[pre]);
            fflush(stdout);
        }
    mg_attn_w64_flag_counter(nullptr);
    mg_attn_w64_debug(0);
    mg_attn_set_variant(0);
}

static void test_small() {
    {  // sinusoid
        std::vector<int64_t> t = {999, 500, 3};
        Dev<int64_t> dt(t);
        Dev<float> o(3 * 256);
        int rc = mg_sinusoid_embed(dt.p, 0, 3, 256, o.p, 0);
        CK(hipDeviceSynchronize());
        auto h = o.host();
        double err = rc ? 1e9 : 0;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 128; ++j) {
                double a = (double)t[i] * pow(10000.0, -(double)j / 128);
                err = fmax(err, fabs(h[i * 256 + j] - cos(a)));
                err = fmax(err, fabs(h[i * 256 + 128 + j] - sin(a)));
            }
        report("sinusoid_embed", err, 1e-6);
    }
    {  // gemv
        const int N = 777, K = 5120;
        auto W = randf((size_t)N * K, 0.02f), b = randf(N), x = randf(K);
        Dev<float> dW(W), db(b), dx(x), dy(N);
        for (int silu = 0; silu < 2; ++silu) {
            int rc = mg_gemv_f32(dW.p, db.p, dx.p, dy.p, N, K, silu, 0);
            CK(hipDeviceSynchronize());
            auto y = dy.host();
            double err = rc ? 1e9 : 0;
            for (int n = 0; n < N; ++n) {
                double a = b[n];
                for (int kx = 0; kx < K; ++kx) { double xv = x[kx]; if (silu) xv = xv / (1 + exp(-xv)); a += (double)W[(size_t)n * K + kx] * xv; }
                err = fmax(err, fabs(y[n] - a));
            }
            report(silu ? "gemv_f32 silu_in" : "gemv_f32", err, 2e-5);
        }
    }
    {  // add_rows
        const int R = 12, D = 128;
        auto a = randf(R * D), b = randf(6 * D);
        Dev<float> da(a), db(b), dout(R * D);
        int rc = mg_add_rows_f32(da.p, db.p, dout.p, R, D, 6, 0);
        CK(hipDeviceSynchronize());
        auto o = dout.host();
        double err = rc ? 1e9 : 0;
        for (int r = 0; r < R; ++r) for (int c = 0; c < D; ++c) err = fmax(err, fabs(o[r * D + c] - (a[r * D + c] + b[(r % 6) * D + c])));
        report("add_rows_f32", err, 0);
    }
    {  // head gemm
        const int64_t M = 333; const int N = 64, K = 5120;
        auto x = randf((size_t)M * K), W = randf((size_t)N * K, 0.02f), b = randf(N);
        Dev<float> dx(x), dW(W), db(b), dout((size_t)M * N);
        int rc = mg_head_gemm_f32(dx.p, K, dW.p, db.p, dout.p, M, N, K, 0);
        CK(hipDeviceSynchronize());
        auto o = dout.host();
        double err = rc ? 1e9 : 0;
        for (int64_t m = 0; m < M; ++m) for (int n = 0; n < N; ++n) {
            double a = b[n];
            for (int kx = 0; kx < K; ++kx) a += (double)x[(size_t)m * K + kx] * W[(size_t)n * K + kx];
            err = fmax(err, fabs(o[(size_t)m * N + n] - a));
        }
        report("head_gemm_f32 M333 N64 K5120", err, 1e-4);
    }
    {  // patchify / unpatchify
        const int C = 16, F = 3, H = 8, W = 12, ph = 2, pw = 2;
        auto lat = randf((size_t)C * F * H * W);
        const int Hg = H / ph, Wg = W / pw, Kd = C * ph * pw; const int64_t L = (int64_t)F * Hg * Wg;
        Dev<float> dl(lat); Dev<uint16_t> dt((size_t)L * Kd);
        int rc = mg_patchify_bf16(dl.p, C, F, H, W, ph, pw, dt.p, Kd, 0);
        CK(hipDeviceSynchronize());
        auto t = dt.host();
        double err = rc ? 1e9 : 0;
        for (int64_t tok = 0; tok < L; ++tok) for (int kx = 0; kx < Kd; ++kx) {
            int c = kx / (ph * pw), ii = (kx % (ph * pw)) / pw, jj = kx % pw;
            int f = tok / (Hg * Wg), hg = (tok % (Hg * Wg)) / Wg, wg = tok % Wg;
            float ref = rbf(lat[(((size_t)c * F + f) * H + hg * ph + ii) * W + wg * pw + jj]);
            err = fmax(err, fabs(bf2f(t[(size_t)tok * Kd + kx]) - ref));
        }
        report("patchify_bf16", err, 0);
        auto tokf = randf((size_t)L * Kd);
        Dev<float> dtf(tokf), dlat((size_t)C * F * H * W);
        rc = mg_unpatchify_f32(dtf.p, Kd, C, F, Hg, Wg, ph, pw, dlat.p, 0);
        CK(hipDeviceSynchronize());
        auto lo = dlat.host();
        err = rc ? 1e9 : 0;
        for (int c = 0; c < C; ++c) for (int f = 0; f < F; ++f) for (int h = 0; h < H; ++h) for (int w = 0; w < W; ++w) {
            int64_t tok = ((int64_t)f * Hg + h / ph) * Wg + w / pw;
            float ref = tokf[(size_t)tok * Kd + ((h % ph) * pw + (w % pw)) * C + c];
            err = fmax(err, fabs(lo[(((size_t)c * F + f) * H + h) * W + w] - ref));
        }
        report("unpatchify_f32", err, 0);
    }
    {  // lincomb
        const int64_t n = 100003;
        auto a = randf(n), b = randf(n), c = randf(n);
        Dev<float> da(a), db(b), dc(c), dout(n);
        int rc = mg_lincomb4_f32(dout.p, n, da.p, 0.5f, db.p, -2.f, nullptr, 0.f, dc.p, 3.f, 0);
        CK(hipDeviceSynchronize());
        auto o = dout.host();
        double err = rc ? 1e9 : 0;
        for (int64_t i = 0; i < n; ++i) err = fmax(err, fabs(o[i] - (0.5 * a[i] - 2.0 * b[i] + 3.0 * c[i])));
        report("lincomb4_f32", err, 2e-6);
    }
}

static void bench_elementwise(int64_t L, int dim) {
    Dev<float> x((size_t)L * dim), sc(dim), sh(dim);
    Dev<uint16_t> o((size_t)L * dim), qkv((size_t)L * dim * 3), q2((size_t)L * dim);
    Dev<float> w(dim);
    x.zero(); sc.zero(); sh.zero(); w.zero(); qkv.zero();
    float ms = time_ms([&] { mg_ln_modulate(x.p, dim, L, dim, sc.p, sh.p, 1, 1e-6f, 0, o.p, 0, dim, 0); }, 5);
    printf("    ln_modulate L=%lld: %.3f ms  %.2f TB/s\n", (long long)L, ms, (double)L * dim * 6 / (ms * 1e-3) / 1e12);
    const int c = 64, c1 = 21, c0 = 22;
    const int F = 21, H = 45, W = 80;
    Dev<float> cs((size_t)2 * (F * c0 + H * c1 + W * c1));
    cs.zero();
    (void)c;
    ms = time_ms([&] { mg_rmsnorm_rope_bf16(qkv.p, 3 * dim, q2.p, dim, L, dim, w.p, 1e-6f, 128, cs.p, F, H, W, 0, 1.0f, 0); }, 5);
    printf("    rmsnorm_rope L=%lld: %.3f ms  %.2f TB/s\n", (long long)L, ms, (double)L * dim * 4 / (ms * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
    const bool full = argc > 1 && !strcmp(argv[1], "full");
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s  CUs=%d  %s  abi=%d\n", prop.name, prop.multiProcessorCount, mg_version(), mg_abi_version());
    if (argc > 1 && !strcmp(argv[1], "attn")) {  // quick perf loop on the dominant kernel
        for (int variant : {0, 3}) {
            printf("== attention variant %d (%s) ==\n", variant, variant ? "w64, round-2 kernel" : "m16");
            mg_attn_set_variant(variant);
            for (int pre = 0; pre < 2; ++pre) {
                test_attn(300, 300, 2, 0, false, pre);
                test_attn(64, 64, 1, 0, false, pre);
                test_attn(700, 512, 3, 0, false, pre);
                test_attn(1000, 77, 1, 0, false, pre);
                test_attn(75600, 75600, 8, 24, true, pre);
                test_attn(75600, 512, 40, 24, true, pre);
            }
        }
        mg_attn_set_variant(0);
        printf("%s: %d failure(s)\n", n_fail ? "SELFTEST FAILED" : "SELFTEST OK", n_fail);
        return n_fail ? 1 : 0;
    }
    if (argc > 1 && !strcmp(argv[1], "attnab")) {   // attnab L heads data rounds v1 v2 ...
        const int64_t L = argc > 2 ? atoll(argv[2]) : 131040;
        const int heads = argc > 3 ? atoi(argv[3]) : 8;
        const int data = argc > 4 ? atoi(argv[4]) : 0;
        const int rounds = argc > 5 ? atoi(argv[5]) : 2;
        std::vector<int> vars;
        for (int i = 6; i < argc; ++i) vars.push_back(atoi(argv[i]));
        if (vars.empty()) vars = {0, 3};
        attn_ab(L, heads, data, vars, rounds);
        printf("%s: %d failure(s)\n", n_fail ? "SELFTEST FAILED" : "SELFTEST OK", n_fail);
        return n_fail ? 1 : 0;
    }
    if (argc > 1 && !strcmp(argv[1], "w64dbg")) {  // pipelined pass only (flagged blocks are NOT redone)
        mg_attn_set_variant(argc > 3 ? atoi(argv[3]) : 0);
        mg_attn_w64_debug(argc > 2 ? atoi(argv[2]) : 1);
        test_attn(300, 300, 2, 0, false);
        test_attn(300, 256, 2, 0, false);
        test_attn(300, 192, 1, 0, false);
        test_attn(300, 320, 1, 0, false);
        test_attn(300, 257, 1, 0, false);
        test_attn(700, 512, 3, 0, false);
        test_attn(75600, 75600, 8, 24, true);
        test_attn(75600, 75584, 8, 24, true);
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "w64prof")) {  // s_memtime breakdown of the w64 hot loop
        unsigned long long* buf;
        CK(hipMalloc(&buf, (16 + 1024 + 8) * 8));
        CK(hipMemset(buf, 0, (16 + 1024 + 8) * 8));
        mg_attn_set_variant(argc > 4 ? atoi(argv[4]) : 0);
        mg_attn_w64_profile(buf);                // (both kernels share the hook)
        if (argc > 6) mg_attn_w64_debug(atoi(argv[6]));
        test_attn(argc > 7 ? atoll(argv[7]) : 75600, argc > 2 ? atoll(argv[2]) : 75584, argc > 3 ? atoi(argv[3]) : 8, 8, true, argc > 5 ? atoi(argv[5]) : 1);
        unsigned long long h[16];
        CK(hipMemcpy(h, buf, sizeof h, hipMemcpyDeviceToHost));
        for (int w = 0; w < 4; ++w) {
            const double n = (double)h[w * 4 + 3];
            printf("wave %d: iterations %.0f  fence %.0f  step A %.0f  step B %.0f  (cycles per tile, 64 MFMAs)\n", w, n, h[w * 4] / n,
                   h[w * 4 + 1] / n, h[w * 4 + 2] / n);
        }
        {   // the last launch's {start, end} of every workgroup on the shared 100 MHz counter: what the static item partition loses to its tail
            std::vector<unsigned long long> se(1024);
            CK(hipMemcpy(se.data(), buf + 16, 1024 * 8, hipMemcpyDeviceToHost));
            unsigned long long t0 = ~0ull, t1 = 0;
            int nw = 0;
            for (int b = 0; b < 512 && se[2 * b + 1]; ++b) t0 = std::min(t0, se[2 * b]), t1 = std::max(t1, se[2 * b + 1]), ++nw;
            if (nw) {
                double mean_end = 0, xe[8] = {0}, xmax[8] = {0}, xmin[8];
                int xn[8] = {0};
                for (int x = 0; x < 8; ++x) xmin[x] = 1e30;
                for (int b = 0; b < nw; ++b) {
                    const double e = (double)(se[2 * b + 1] - t0) / 100.0;      // microseconds
                    mean_end += e, xe[b & 7] += e, xn[b & 7]++, xmax[b & 7] = std::max(xmax[b & 7], e), xmin[b & 7] = std::min(xmin[b & 7], e);
                }
                printf("workgroups %d: first start -> last end %.1f us; mean end %.1f us (a balanced kernel would end there: %.2f %% lost to the tail)\n", nw,
                       (double)(t1 - t0) / 100.0, mean_end / nw, 100.0 * (1.0 - mean_end / nw / ((double)(t1 - t0) / 100.0)));
                for (int x = 0; x < 8; ++x)
                    if (xn[x]) printf("  XCD %d: first end %.1f  mean end %.1f  last end %.1f us\n", x, xmin[x], xe[x] / xn[x], xmax[x]);
            }
        }
        {   // m16: wave 0's per-item phases (s_memtime cycles), summed over workgroups and launches
            unsigned long long ph[8];
            CK(hipMemcpy(ph, buf + 1040, sizeof ph, hipMemcpyDeviceToHost));
            if (ph[6]) {
                const double n = (double)ph[6];
                printf("per item (%.0f items), cycles: Q + first K / V tiles landed %.0f | first tile: S, reference, softmax, step 1 %.0f | wait for the refills %.0f | steady loop %.0f | "
                       "drain + row sums %.0f | normalise + store O, stores drained %.0f (the last store issued after %.0f) | sum %.0f\n",
                       n, ph[0] / n, ph[1] / n, ph[2] / n, ph[3] / n, ph[4] / n, ph[5] / n, ph[7] / n, (ph[0] + ph[1] + ph[2] + ph[3] + ph[4] + ph[5]) / n);
            }
        }
        mg_attn_w64_profile(nullptr);
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "attnpmc")) {  // attnpmc L heads: two launches of the pre-scaled entry on constant operands, nothing
        // else — what bench.py runs under `rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE` to read the kernel's HBM-side traffic
        const int64_t L = argc > 2 ? atoll(argv[2]) : 131040;
        const int heads = argc > 3 ? atoi(argv[3]) : 40;
        const int64_t ld = (int64_t)heads * 128, npk = (int64_t)heads * ((L + 63) / 64) * 8192;
        Dev<uint16_t> dq((size_t)L * ld), dkp((size_t)npk), dvp((size_t)npk), dout((size_t)L * ld);
        CK(hipMemset(dq.p, 0x3c, dq.n * 2));     // bf16 0x3c3c = 0.0115: traffic does not depend on the values
        CK(hipMemset(dkp.p, 0x3c, dkp.n * 2));
        CK(hipMemset(dvp.p, 0x3c, dvp.n * 2));
        int rc = 0;
        for (int i = 0; i < 2; ++i) rc |= mg_attn_fwd_bf16_hd128_prescaled(dq.p, ld, dkp.p, dvp.p, dout.p, ld, nullptr, L, L, heads, 0, attn_ws(), 0);
        CK(hipDeviceSynchronize());
        printf("attnpmc L=%lld heads=%d rc=%d\n", (long long)L, heads, rc);
        return rc ? 1 : 0;
    }
    if (argc > 1 && !strcmp(argv[1], "attn1")) {  // single big launch set, for rocprofv3 --pmc passes
        mg_attn_set_variant(argc > 2 ? atoi(argv[2]) : 0);
        test_attn(argc > 3 ? atoll(argv[3]) : 75600, argc > 3 ? atoll(argv[3]) : 75600, 8, 8, true, 1);
        return n_fail ? 1 : 0;
    }
    if (argc > 1 && !strcmp(argv[1], "gemmshapes")) {   // the five GEMMs of a DiT block at 720p, with their epilogues
        mg_gemm_set_variant(argc > 2 ? atoi(argv[2]) : 8);
        const int64_t Mg = argc > 3 ? atoll(argv[3]) : 75600;   // 131040 = the 1920x832x81f token count
        test_gemm(Mg, 15360, 5120, 0, 128, true);     // q|k|v
        test_gemm(Mg, 5120, 5120, 2, 128, true);      // self-attention o (+ gate, residual)
        test_gemm(Mg, 5120, 5120, 0, 128, true);      // cross-attention q
        test_gemm(Mg, 13824, 5120, 1, 128, true);     // ffn.0 + GELU
        test_gemm(Mg, 5120, 13824, 2, 128, true);     // ffn.2 (+ gate, residual)
        return n_fail ? 1 : 0;
    }
    if (argc > 1 && !strcmp(argv[1], "powerloop")) {   // powerloop gemm <variant> <secs> | attn 0 <secs>: ONE kernel back to back on random operands
        // (tools/r04_gpu_w.sh samples rocm-smi power / clocks meanwhile: are the hot kernels at the power cap?)
        const bool is_gemm = !strcmp(argv[2], "gemm");
        const int var = atoi(argv[3]);
        const double secs = argc > 4 ? atof(argv[4]) : 8.0;
        const int64_t M = 131040;
        auto blk = randbf((size_t)32 << 20);                 // 64 MiB of random bf16, tiled over the big operands
        auto fill = [&](uint16_t* d, size_t n) {
            for (size_t o = 0; o < n; o += blk.size())
                CK(hipMemcpy(d + o, blk.data(), std::min(blk.size(), n - o) * 2, hipMemcpyHostToDevice));
        };
        double ms_sum = 0;
        long launches = 0;
        if (is_gemm) {
            // powerloop gemm <variant> <secs> [N] [K] [epilogue] [data]: data 1 = the A operand all zero (same instruction stream, no switching in the multipliers)
            const int N = argc > 5 ? atoi(argv[5]) : 5120, K = argc > 6 ? atoi(argv[6]) : 5120;
            const int epi = argc > 7 ? atoi(argv[7]) : MG_EPI_GATE_RESID_F32, data = argc > 8 ? atoi(argv[8]) : 0;
            Dev<uint16_t> dA((size_t)M * K), dW(randbf((size_t)N * K, 0.05f));
            if (data) CK(hipMemset(dA.p, 0, dA.n * 2)); else fill(dA.p, dA.n);
            Dev<float> db(randf(N)), dg(randf(N)), of((size_t)M * N);
            CK(hipMemset(of.p, 0, of.n * 4));
            mg_gemm_set_variant(var);
            while (ms_sum < secs * 1e3) {
                ms_sum += 20 * time_ms([&] { mg_gemm_bf16(dA.p, K, dW.p, K, db.p, M, N, K, epi, of.p, N, dg.p, 0); }, 20);
                launches += 20;
            }
            printf("powerloop gemm variant %d N %d K %d epi %d data %d: %ld launches, %.3f ms each = %.1f TFLOP/s\n", var, N, K, epi, data, launches, ms_sum / launches,
                   2.0 * M * N * K / (ms_sum / launches * 1e-3) / 1e12);
        } else {
            // powerloop attn <debug flags> <secs> [heads] [data] [kernel]: debug flags as mg_attn_w64_debug (2 k = filler placement k, 16 = static partition);
            // data 0 = random bf16 (what a real launch sees), 1 = all zero, 2 = the constant 0x3c3c (no operand bit ever toggles): the same instruction stream
            // at a different switching activity — what the package-power limit costs; kernel 3 = the round-2 kernel (w64)
            const int heads = argc > 5 ? atoi(argv[5]) : 8;
            const int data = argc > 6 ? atoi(argv[6]) : 0;
            if (argc > 7) mg_attn_set_variant(atoi(argv[7]));
            mg_attn_w64_debug(var);
            const int64_t ld = heads * 128, npk = (int64_t)heads * ((M + 63) / 64) * 8192;
            Dev<uint16_t> dq((size_t)M * ld), dkp((size_t)npk), dvp((size_t)npk), dout((size_t)M * ld);
            if (data == 0) { fill(dq.p, dq.n); fill(dkp.p, dkp.n); fill(dvp.p, dvp.n); }
            else {
                const int byte = data == 1 ? 0 : 0x3c;
                CK(hipMemset(dq.p, byte, dq.n * 2)); CK(hipMemset(dkp.p, byte, dkp.n * 2)); CK(hipMemset(dvp.p, byte, dvp.n * 2));
            }
            while (ms_sum < secs * 1e3) {
                ms_sum += 4 * time_ms([&] { mg_attn_fwd_bf16_hd128_prescaled(dq.p, ld, dkp.p, dvp.p, dout.p, ld, nullptr, M, M, heads, 0, attn_ws(), 0); }, 4);
                launches += 4;
            }
            printf("powerloop attention flags %d heads %d data %d: %ld launches, %.3f ms each = %.1f TFLOP/s\n", var, heads, data, launches, ms_sum / launches,
                   4.0 * M * M * 128 * heads / (ms_sum / launches * 1e-3) / 1e12);
        }
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "gemmdiff")) {   // gemmdiff variant M N K epi: every element against variant 8, mismatches by position in the wave's block
        const int var = atoi(argv[2]);
        const int64_t M = atoll(argv[3]);
        const int N = atoi(argv[4]), K = atoi(argv[5]), epi = atoi(argv[6]);
        auto A = randbf((size_t)M * K), Wt = randbf((size_t)N * K, 0.05f);
        auto bias = randf(N), gate = randf(N);
        auto r0 = randf((size_t)M * N);
        Dev<uint16_t> dA(A), dW(Wt);
        Dev<float> db(bias), dg(gate), o1(r0), o2(r0);
        mg_gemm_set_variant(8);
        int rc = mg_gemm_bf16(dA.p, K, dW.p, K, db.p, M, N, K, epi, o1.p, N, dg.p, 0);
        mg_gemm_set_variant(var);
        rc |= mg_gemm_bf16(dA.p, K, dW.p, K, db.p, M, N, K, epi, o2.p, N, dg.p, 0);
        CK(hipDeviceSynchronize());
        auto h1 = o1.host(), h2 = o2.host();
        long bad = 0, by_c[8] = {0}, by_e[4] = {0}, by_j[8] = {0}, by_G[4] = {0}, by_r[16] = {0}, by_w[4] = {0};
        for (int64_t m = 0; m < M; ++m)
            for (int n = 0; n < N; ++n)
                if (h1[(size_t)m * N + n] != h2[(size_t)m * N + n]) {
                    if (bad < 8) printf("  (m %lld, n %d): variant 8 %.6f  variant %d %.6f  resid %.6f\n", (long long)m, n, h1[(size_t)m * N + n], var, h2[(size_t)m * N + n], r0[(size_t)m * N + n]);
                    ++bad;
                    const int mm = (int)(m & 127), nn = n & 127;
                    ++by_c[nn >> 4], ++by_e[nn & 3], ++by_G[(nn >> 2) & 3], ++by_j[mm >> 4], ++by_r[mm & 15], ++by_w[((m >> 7) & 1) * 2 + ((n >> 7) & 1)];
                }
        printf("rc %d: %ld of %lld elements differ\n", rc, bad, (long long)(M * N));
        auto show = [](const char* nm, long* v, int k) { printf("  by %s:", nm); for (int i = 0; i < k; ++i) printf(" %ld", v[i]); printf("\n"); };
        show("feature block c", by_c, 8); show("element e", by_e, 4); show("G", by_G, 4); show("token block j", by_j, 8); show("r16", by_r, 16); show("wave", by_w, 4);
        return bad ? 1 : 0;
    }
    if (argc > 1 && !strcmp(argv[1], "gemmv")) {   // gemmv variant M N K [epi]: one shape, every sample row checked
        mg_gemm_set_variant(atoi(argv[2]));
        test_gemm(atoll(argv[3]), atoi(argv[4]), atoi(argv[5]), argc > 6 ? atoi(argv[6]) : 0, 512, false);
        return n_fail ? 1 : 0;
    }
    if (argc > 1 && !strcmp(argv[1], "gemmab")) {   // gemmab M rounds v1 v2 ...: the three residual / one store GEMM of a block, variants alternating
        const int64_t Mg = argc > 2 ? atoll(argv[2]) : 131040;
        const int rounds = argc > 3 ? atoi(argv[3]) : 2;
        std::vector<int> vars;
        for (int i = 4; i < argc; ++i) vars.push_back(atoi(argv[i]));
        if (vars.empty()) vars = {8, 7};
        gemm_ab(Mg, 5120, 5120, 2, vars, rounds);      // self-attention o (+ gate, residual)
        gemm_ab(Mg, 5120, 13824, 2, vars, rounds);     // ffn.2 (+ gate, residual)
        gemm_ab(Mg, 5120, 5120, 0, vars, rounds);      // cross-attention q: store epilogue, the variants must tie
        printf("%s: %d failure(s)\n", n_fail ? "SELFTEST FAILED" : "SELFTEST OK", n_fail);
        return n_fail ? 1 : 0;
    }
    if (argc > 1 && !strcmp(argv[1], "gemmab1")) {   // gemmab1 M N K epi rounds v1 v2 ...: one shape, variants alternating on one set of operands
        std::vector<int> vars;
        for (int i = 7; i < argc; ++i) vars.push_back(atoi(argv[i]));
        gemm_ab(atoll(argv[2]), atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), vars, atoi(argv[6]));
        return n_fail ? 1 : 0;
    }
    if (argc > 1 && !strcmp(argv[1], "gemmprof")) {  // s_memtime breakdown of the 256x128 GEMM k-loop
        unsigned long long* buf;
        CK(hipMalloc(&buf, (64 + 1024) * 8));
        CK(hipMemset(buf, 0, (64 + 1024) * 8));
        const int gv = argc > 2 ? atoi(argv[2]) : 7;
        mg_gemm_set_variant(gv);
        if (gv >= 7) mg_gemm5_debug_profile(buf); else mg_gemm_debug_profile(buf);
        test_gemm(argc > 3 ? atoll(argv[3]) : 75600, argc > 4 ? atoi(argv[4]) : 5120, argc > 5 ? atoi(argv[5]) : 5120, argc > 6 ? atoi(argv[6]) : 0, 64, true);
        unsigned long long h[64];
        CK(hipMemcpy(h, buf, sizeof h, hipMemcpyDeviceToHost));
        if (gv == 12 || gv >= 200) {      // variant 12: 4 waves x {BAR1 wait, BAR2 wait, BAR3 wait, whole body, bodies}; the last k-tile of a tile is not a body
            for (int w = 0; w < 4; ++w) {
                const double n = (double)h[w * 5 + 4];
                printf("wave %d: bodies %.0f  wait at the lgkmcnt barriers %.0f  at the vmcnt barrier %.0f  whole body %.0f (cycles per k-tile)\n", w, n,
                       h[w * 5] / n, h[w * 5 + 2] / n, h[w * 5 + 3] / n);
                const double tiles = (double)h[32 + w * 3 + 2], tot = (double)h[32 + w * 3], epi = (double)h[32 + w * 3 + 1];
                printf("        per tile: %.0f cycles = bodies %.0f + last k-tile / epilogue / tile prologue %.0f + rest %.0f   (%.0f tiles)\n", tot / tiles,
                       (double)h[w * 5 + 3] / tiles, epi / tiles, (tot - (double)h[w * 5 + 3] - epi) / tiles, tiles);
                printf("        of it: last k-tile %.0f  vmcnt(0) + barrier %.0f  epilogue %.0f  tile prologue %.0f\n", h[44 + w * 4] / tiles, h[44 + w * 4 + 1] / tiles,
                       h[44 + w * 4 + 2] / tiles, h[44 + w * 4 + 3] / tiles);
            }
            // the last launch's {start, end} of every workgroup on the shared 100 MHz counter: is the static tile assignment balanced?
            std::vector<unsigned long long> se(1024);
            CK(hipMemcpy(se.data(), buf + 64, 1024 * 8, hipMemcpyDeviceToHost));
            unsigned long long t0 = ~0ull, t1 = 0;
            int nw = 0;
            for (int b = 0; b < 512 && se[2 * b + 1]; ++b) t0 = std::min(t0, se[2 * b]), t1 = std::max(t1, se[2 * b + 1]), ++nw;
            if (nw) {
                double mean_end = 0, xe[8] = {0}, xs[8] = {0}, xmax[8] = {0};
                int xn[8] = {0};
                for (int b = 0; b < nw; ++b) {
                    const double e = (double)(se[2 * b + 1] - t0) / 100.0, st = (double)(se[2 * b] - t0) / 100.0;      // microseconds
                    mean_end += e, xe[b & 7] += e, xs[b & 7] += st, xn[b & 7]++, xmax[b & 7] = std::max(xmax[b & 7], e);
                }
                printf("workgroups %d: first start -> last end %.1f us; mean end %.1f us (a balanced kernel would end there: %.1f %% lost to the tail)\n", nw,
                       (double)(t1 - t0) / 100.0, mean_end / nw, 100.0 * (1.0 - mean_end / nw / ((double)(t1 - t0) / 100.0)));
                for (int x = 0; x < 8; ++x)
                    if (xn[x]) printf("  XCD %d: mean start %.1f  mean end %.1f  last end %.1f us\n", x, xs[x] / xn[x], xe[x] / xn[x], xmax[x]);
            }
        } else if (gv == 11 || gv >= 110) {      // variant 11: 4 waves x {barrier, to the first MFMA, k-step 0, k-step 1, k-tiles}
            for (int w = 0; w < 4; ++w) {
                const double n = (double)h[w * 5 + 4];
                printf("wave %d: k-tiles %.0f  vmcnt+barrier %.0f  to first MFMA %.0f  k-step 0 %.0f  k-step 1 %.0f  sum %.0f (cycles per k-tile)\n", w, n,
                       h[w * 5] / n, h[w * 5 + 1] / n, h[w * 5 + 2] / n, h[w * 5 + 3] / n, (h[w * 5] + h[w * 5 + 1] + h[w * 5 + 2] + h[w * 5 + 3]) / n);
                // 32 + 3w: {whole kernel, tail + epilogue, tiles}, summed over the workgroups
                const double tiles = (double)h[32 + w * 3 + 2], tot = (double)h[32 + w * 3], epi = (double)h[32 + w * 3 + 1];
                const double loop = (double)(h[w * 5] + h[w * 5 + 1] + h[w * 5 + 2] + h[w * 5 + 3]);
                printf("        per tile: %.0f cycles = k-loop %.0f + tail/epilogue %.0f + rest %.0f   (%.0f tiles)\n", tot / tiles, loop / tiles,
                       epi / tiles, (tot - loop - epi) / tiles, tiles);
            }
        } else if (gv >= 8) {      // ping-pong kernels: waves 0-3 = group X, 4-7 = group Y; per PHASE (4 phases = one k-tile of 64 MFMAs per wave)
            for (int w = 0; w < 8; ++w) {
                const double n = (double)h[w * 5 + 4];
                printf("wave %d: phases %.0f  load part %.0f  barrier-1 wait %.0f  MFMA part %.0f  barrier-2 wait %.0f  (cycles per phase; x4 per k-tile)\n",
                       w, n, h[w * 5] / n, h[w * 5 + 1] / n, h[w * 5 + 2] / n, h[w * 5 + 3] / n);
            }
        } else
        for (int w = 0; w < 8; ++w) {
            const double n = (double)h[w * 4 + 3];
            printf("wave %d: k-tiles %.0f  wait+barrier %.0f  stage issue %.0f  MFMA segment %.0f  (cycles per k-tile)\n", w, n,
                   h[w * 4] / n, h[w * 4 + 1] / n, h[w * 4 + 2] / n);
        }
        mg_gemm_debug_profile(nullptr);
        mg_gemm5_debug_profile(nullptr);
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "gemm1")) {
        test_gemm(75600, 5120, 5120, 0, 64, true);
        return n_fail ? 1 : 0;
    }
    if (argc > 1 && !strcmp(argv[1], "gemm")) {
        for (int variant : {1, 2, 7, 8}) {
            printf("== gemm variant %d ==\n", variant);
            mg_gemm_set_variant(variant);
            for (int epi = 0; epi < 4; ++epi) test_gemm(300, 256, 128, epi, 0, false);
            test_gemm(129, 200, 192, 2, 0, false);
            test_gemm(257, 128, 64, 0, 0, false);
            test_gemm(1000, 1280, 1024, 2, 0, false);
            test_gemm(515, 64, 5120, 3, 0, false);
            test_gemm(75600, 5120, 5120, 0, 128, true);
            test_gemm(75600, 13824, 5120, 1, 128, true);
            test_gemm(75600, 5120, 13824, 2, 128, true);
            test_gemm(8192, 8192, 8192, 0, 128, true);
        }
        printf("%s: %d failure(s)\n", n_fail ? "SELFTEST FAILED" : "SELFTEST OK", n_fail);
        return n_fail ? 1 : 0;
    }

    test_small();
    test_ln(37, 5120, 1, 0, 0);
    test_ln(37, 5120, 0, 0, 1);
    test_ln(5, 128, 1, 1, 0);
    test_ln(9, 8192, 1, 0, 1);
    test_rmsnorm_rope(60, 5120, 128, 2, 5, 5, 0, true);   // rows 50..59 are padding tokens
    test_rmsnorm_rope(25, 5120, 128, 2, 5, 5, 25, true);  // SP rank slice
    test_rmsnorm_rope(17, 128, 32, 1, 4, 4, 0, true);
    test_rmsnorm_rope(33, 5120, 128, 1, 1, 1, 0, false);
    test_pack(300, 2);
    test_pack(64, 1);

    for (int epi = 0; epi < 4; ++epi) test_gemm(300, 256, 128, epi, 0, false);
    test_gemm(128, 128, 64, 0, 0, false);
    test_gemm(129, 200, 192, 0, 0, false);   // M and N edges
    test_gemm(77, 64, 5120, 3, 0, false);    // head-like narrow N
    test_gemm(1000, 1280, 1024, 2, 0, false);  // many tiles -> raster / XCD remap

    test_attn(300, 300, 2, 0, false, 0);
    test_attn(300, 300, 2, 0, false, 1);
    test_attn(700, 512, 3, 0, false, 1);   // cross-attention shape
    test_attn(64, 64, 1, 0, false, 0);
    test_attn(1000, 77, 1, 0, false, 0);   // short, ragged key length

    if (full) {
        printf("---- 14B / 720p shapes (L=75600, d=5120, ffn=13824) ----\n");
        const int64_t L = 75600;
        bench_elementwise(L, 5120);
        test_gemm(L, 5120, 5120, 0, 256, true);
        test_gemm(L, 15360, 5120, 0, 256, true);
        test_gemm(L, 13824, 5120, 1, 256, true);
        test_gemm(L, 5120, 13824, 2, 256, true);
        test_gemm(4096, 4096, 4096, 0, 256, true);
        test_gemm(8192, 8192, 8192, 0, 256, true);
        test_attn(L, 512, 40, 48, true, 1);
        test_attn(8192, 8192, 40, 48, true, 1);
        test_attn(L, L, 40, 40, true, 1);
        test_attn(L, L, 40, 24, true, 0);      // the general entry (one v_mul more per score)
    }
    printf("%s: %d failure(s)\n", n_fail ? "SELFTEST FAILED" : "SELFTEST OK", n_fail);
    return n_fail ? 1 : 0;
}
