// Lab (GPU box only): the persistent decode chain (tools/gemv_chain_lab.h) against the product's four launches
// (qeft_decode_linear x 4 through libqeft_hip.so) on the operands of one Llama-2-7B layer: o_proj -> gate|up -> down_proj -> q|k|v.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I qeft_amd/csrc -I include -I tools tools/chain_lab.hip -L qeft_amd/lib -lqeft_hip -Wl,-rpath,$PWD/qeft_amd/lib -o build/chain_lab
// Prints: correctness of the chain against the four launches (h32, q|k|v), us per layer-chain both ways (interleaved rounds,
// 8 weight sets cycled so that nothing is served from L2 / MALL), and the in-kernel timeline of the chain's edges.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <algorithm>
#ifndef LAB_NW
#define LAB_NW 8
#endif
#ifndef LAB_D
#define LAB_D 11
#endif
#include "gemv_chain_lab.h"
#include "qeft_hip.h"

namespace qeft { thread_local const char* g_last_variant = ""; }
using namespace qeft;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#define QK(x) do { int e = (x); if (e != 0) { printf("qeft error %d at %d\n", e, __LINE__); exit(1);} } while (0)

__global__ void fill_random(uint32_t* p, size_t n, uint32_t seed, uint32_t andmask, uint32_t ormask) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t v = (uint32_t)i * 2654435761u ^ seed;
    v ^= v >> 16; v *= 0x85ebca6bu; v ^= v >> 13; v *= 0xc2b2ae35u; v ^= v >> 16;
    p[i] = (v & andmask) | ormask;
}
static void* dalloc(size_t bytes, uint32_t seed, uint32_t andmask = 0xffffffffu, uint32_t ormask = 0) {
    void* p; CK(hipMalloc(&p, (bytes + 255) / 256 * 256));
    size_t n = (bytes + 3) / 4;
    hipLaunchKernelGGL(fill_random, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (uint32_t*)p, n, seed, andmask, ormask);
    return p;
}
struct Lin { void *qw, *szp, *ow; int n, k; };
static Lin make_lin(int n, int k, uint32_t seed) {
    Lin l; l.n = n; l.k = k;
    l.qw = dalloc((size_t)n * k / 2, seed * 7 + 1);
    // scale ~ 2^-8 .. 2^-7 (fp16 0x1c00 | mantissa), scaled zero ~ -(0.03..0.06) (0xa800 | mantissa): weights ~ +-0.03
    l.szp = dalloc((size_t)n * (k / 128) * 4, seed * 7 + 2, 0x03ff03ffu, 0xa8001c00u);
    l.ow = dalloc((size_t)n * 128 * 2, seed * 7 + 3, 0x83ff83ffu, 0x20002000u);     // +-(0.008 .. 0.016)
    return l;
}
struct Layer { Lin o, gu, d, qkv; };

static float h2f(uint16_t h) {
    uint32_t s = (h >> 15) & 1, e = (h >> 10) & 31, m = h & 1023, u;
    if (e == 0) { if (!m) u = s << 31; else { e = 1; while (!(m & 1024)) { m <<= 1; --e; } m &= 1023; u = (s << 31) | ((e + 112) << 23) | (m << 13); } }
    else if (e == 31) u = (s << 31) | 0x7f800000u | (m << 13);
    else u = (s << 31) | ((e + 112) << 23) | (m << 13);
    float f; memcpy(&f, &u, 4); return f;
}

template <int NW, int D, bool DBG, bool NOMATH = false>
static void launch_chain(const ChArgs& a, size_t smem, hipStream_t st) {
    auto kern = gemv_chain_kernel<NW, D, DBG, NOMATH>;
    static bool set = false;
    if (!set) { CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); set = true; }
    hipLaunchKernelGGL(kern, dim3(a.nblk), dim3(NW * 64), smem, st, a);
}

int main(int argc, char** argv) {
    const int hidden = 4096, inter = 11008, kv = 4096;
    const int L = 8;
    const float eps = 1e-5f;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int nblk = std::min(256, prop.multiProcessorCount);
    printf("device: %s, %d CUs, chain grid %d blocks x %d threads\n", prop.name, prop.multiProcessorCount, nblk, LAB_NW * 64);
    std::vector<Layer> W(L);
    for (int l = 0; l < L; ++l) {
        W[l].o = make_lin(hidden, hidden, 100 + l);
        W[l].gu = make_lin(2 * inter, hidden, 200 + l);
        W[l].d = make_lin(hidden, inter, 300 + l);
        W[l].qkv = make_lin(hidden + 2 * kv, hidden, 400 + l);
    }
    void* att = dalloc(hidden * 2, 9, 0x83ff83ffu, 0x38003800u);               // +-(0.5 .. 1)
    void* gam1 = dalloc(hidden * 2, 11, 0x03ff03ffu, 0x3c003c00u);             // 1 .. 2
    void* gam2 = dalloc(hidden * 2, 12, 0x03ff03ffu, 0x3c003c00u);
    void* h0 = dalloc(hidden * 4, 13, 0x807fffffu, 0x3f000000u);               // +-(0.5 .. 1)
    float *h32a, *h32b, *ssq; f16 *xn, *act, *qkva, *qkvb;
    CK(hipMalloc(&h32a, hidden * 4)); CK(hipMalloc(&h32b, hidden * 4)); CK(hipMalloc(&ssq, 4096 * 4));
    CK(hipMalloc(&xn, hidden * 2)); CK(hipMalloc(&act, inter * 2)); CK(hipMalloc(&qkva, (hidden + 2 * kv) * 2)); CK(hipMalloc(&qkvb, (hidden + 2 * kv) * 2));
    const int n_ssq = qeft_decode_linear_blocks(hidden);

    // ---- the product's four launches
    auto baseline = [&](const Layer& w, float* h32, f16* qkv, hipStream_t st) {
        QK(qeft_decode_linear(att, w.o.qw, w.o.szp, w.o.ow, nullptr, h32, hidden, hidden, 128, 128, 0, h32, nullptr, 0, eps, gam1, xn, ssq, st));
        QK(qeft_decode_linear(xn, w.gu.qw, w.gu.szp, w.gu.ow, nullptr, act, 2 * inter, hidden, 128, 128, 1, nullptr, ssq, n_ssq, eps, nullptr, nullptr, nullptr, st));
        QK(qeft_decode_linear(act, w.d.qw, w.d.szp, w.d.ow, nullptr, h32, hidden, inter, 128, 128, 0, h32, nullptr, 0, eps, gam2, xn, ssq, st));
        QK(qeft_decode_linear(xn, w.qkv.qw, w.qkv.szp, w.qkv.ow, nullptr, qkv, hidden + 2 * kv, hidden, 128, 128, 0, nullptr, ssq, n_ssq, eps, nullptr, nullptr, nullptr, st));
    };

    // ---- the chain
    constexpr int NW = LAB_NW, D = LAB_D;
    const uint32_t gran_ssq_off = 8192;                       // >= max K / 2
    unsigned long long* gran[3];
    for (auto& g : gran) { CK(hipMalloc(&g, (gran_ssq_off + 512) * 8)); CK(hipMemset(g, 0, (gran_ssq_off + 512) * 8)); }
    uint32_t *epoch, *status; CK(hipMalloc(&epoch, 256)); CK(hipMalloc(&status, 256));
    CK(hipMemset(epoch, 0, 256)); CK(hipMemset(status, 0, 256));
    long long* dbg; CK(hipMalloc(&dbg, (size_t)nblk * CH_MAX_PHASES * 16 * 8)); CK(hipMemset(dbg, 0, (size_t)nblk * CH_MAX_PHASES * 16 * 8));
    auto phase = [&](const Lin& l, int epi, int norm_in, const void* gamma, float* h32, f16* y, const void* x, int store_h) {
        ChPhase p{};
        p.qw = (const uint8_t*)l.qw; p.szp = (const uint8_t*)l.szp; p.ow = (const uint8_t*)l.ow;
        p.gamma_out = (const f16*)gamma; p.h32 = h32; p.y = y; p.x = (const f16*)x;
        p.K = l.k; p.nsets = l.n / 16; p.epi = epi; p.norm_in = norm_in;
        p.sets_q = p.nsets / nblk; p.sets_r = p.nsets % nblk; p.store_h = store_h; p.eps = eps;
        return p;
    };
    auto chain_args = [&](const Layer& w, float* h32, f16* qkv, int nph) {
        ChArgs a{};
        a.ph[0] = phase(w.o, CH_EPI_RESID, 0, gam1, h32, nullptr, att, nph <= 2);
        a.ph[1] = phase(w.gu, CH_EPI_PAIR, 1, nullptr, nullptr, act, nullptr, 0);
        a.ph[2] = phase(w.d, CH_EPI_RESID, 0, nph > 3 ? gam2 : nullptr, h32, nullptr, nullptr, 1);
        a.ph[3] = phase(w.qkv, CH_EPI_STORE, 1, nullptr, nullptr, qkv, nullptr, 0);
        a.nph = nph; a.nblk = nblk;
        for (int e = 0; e < 3; ++e) a.gran[e] = gran[e];
        a.gran[3] = gran[0];
        a.gran_ssq_off = gran_ssq_off; a.epoch = epoch; a.status = status; a.dbg = nullptr;
        a.timeout_ticks = 200000;      // 2 ms
        return a;
    };
    size_t smem;
    {
        uint32_t szmax = 0; int kmax = 0, rscmax = 1;
        ChArgs a = chain_args(W[0], h32b, qkvb, 4);
        for (int p = 0; p < 4; ++p) {
            const int rsc = a.ph[p].sets_q + (a.ph[p].sets_r ? 1 : 0);
            kmax = std::max(kmax, a.ph[p].K); rscmax = std::max(rscmax, rsc);
            szmax = std::max(szmax, (uint32_t)rsc * (uint32_t)v3_sz_bytes(a.ph[p].K >> 7));
            if (rsc > CH_MAX_RSC) { printf("phase %d needs %d row sets per block\n", p, rsc); return 1; }
        }
        const ChLds Ld = ch_lds(kmax, szmax, rscmax, D, NW);
        smem = Ld.total;
        printf("chain LDS: x %u, scales %u, outliers %u, ring %u (NW = %d, D = %d), total %zu bytes\n", Ld.szl - Ld.xs, Ld.owl - Ld.szl, Ld.epl - Ld.owl, Ld.total - Ld.ring, NW, D, smem);
        if (smem > 160 * 1024) { printf("does not fit\n"); return 1; }
    }

    // ---- correctness: the chain against the four launches, on every weight set
    std::vector<float> ha(hidden), hb(hidden); std::vector<uint16_t> qa(hidden + 2 * kv), qb(hidden + 2 * kv);
    int bad = 0;
    for (int l = 0; l < L; ++l) {
        CK(hipMemcpy(h32a, h0, hidden * 4, hipMemcpyDeviceToDevice)); CK(hipMemcpy(h32b, h0, hidden * 4, hipMemcpyDeviceToDevice));
        CK(hipMemset(qkva, 0, qa.size() * 2)); CK(hipMemset(qkvb, 0xff, qb.size() * 2));
        baseline(W[l], h32a, qkva, 0);
        if (getenv("CHAIN_LAB_ZERO")) for (auto& g : gran) CK(hipMemset(g, 0, (gran_ssq_off + 512) * 8));
        launch_chain<NW, D, false>(chain_args(W[l], h32b, qkvb, 4), smem, 0);
        CK(hipDeviceSynchronize());
        uint32_t stv = 0; CK(hipMemcpy(&stv, status, 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(ha.data(), h32a, hidden * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), h32b, hidden * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(qa.data(), qkva, qa.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(qb.data(), qkvb, qb.size() * 2, hipMemcpyDeviceToHost));
        double eh = 0, mh = 0, eq = 0, mq = 0;
        for (int i = 0; i < hidden; ++i) { eh = std::max(eh, (double)fabsf(ha[i] - hb[i])); mh = std::max(mh, (double)fabsf(ha[i])); }
        for (size_t i = 0; i < qa.size(); ++i) { const float a = h2f(qa[i]), b = h2f(qb[i]); eq = std::max(eq, (double)fabsf(a - b)); mq = std::max(mq, (double)fabsf(a)); }
        const bool ok = stv == 0 && eh / mh < 2e-3 && eq / mq < 4e-3 && std::isfinite(eh) && std::isfinite(eq);
        { uint32_t ev = 0; CK(hipMemcpy(&ev, epoch, 4, hipMemcpyDeviceToHost)); printf("[epoch %u] ", ev); }
        printf("set %d: status %#x | h32 max|d| %.3e (max|h| %.3f) | qkv max|d| %.3e (max|q| %.3f) -> %s\n", l, stv, eh, mh, eq, mq, ok ? "OK" : "MISMATCH");
        bad += !ok;
        if (stv) { CK(hipMemset(status, 0, 256)); }
    }
    if (bad) {
        // which phase: chains of 1, 2, 3 phases against the launches' intermediate results
        std::vector<uint16_t> aa(inter), ab(inter);
        f16* act2; CK(hipMalloc(&act2, inter * 2));
        for (int nph = 1; nph <= 3; ++nph) {
            const Layer& w = W[0];
            CK(hipMemcpy(h32a, h0, hidden * 4, hipMemcpyDeviceToDevice)); CK(hipMemcpy(h32b, h0, hidden * 4, hipMemcpyDeviceToDevice));
            QK(qeft_decode_linear(att, w.o.qw, w.o.szp, w.o.ow, nullptr, h32a, hidden, hidden, 128, 128, 0, h32a, nullptr, 0, eps, gam1, xn, ssq, 0));
            if (nph >= 2) QK(qeft_decode_linear(xn, w.gu.qw, w.gu.szp, w.gu.ow, nullptr, act, 2 * inter, hidden, 128, 128, 1, nullptr, ssq, n_ssq, eps, nullptr, nullptr, nullptr, 0));
            if (nph >= 3) QK(qeft_decode_linear(act, w.d.qw, w.d.szp, w.d.ow, nullptr, h32a, hidden, inter, 128, 128, 0, h32a, nullptr, 0, eps, nullptr, nullptr, nullptr, 0));
            ChArgs a = chain_args(w, h32b, qkvb, nph);
            a.ph[1].y = act2;
            if (nph == 1) a.ph[0].store_h = 1;
            if (nph == 3) { a.ph[0].store_h = 0; a.ph[2].gamma_out = nullptr; }
            launch_chain<NW, D, false>(a, smem, 0);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(ha.data(), h32a, hidden * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), h32b, hidden * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(aa.data(), act, inter * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(ab.data(), act2, inter * 2, hipMemcpyDeviceToHost));
            double eh = 0, mh = 0, ea = 0, ma = 0; int nbadh = 0, nbada = 0, firsth = -1, firsta = -1;
            for (int i = 0; i < hidden; ++i) { const double d = fabsf(ha[i] - hb[i]); if (d > 1e-3 * fabsf(ha[i]) + 1e-3) { ++nbadh; if (firsth < 0) firsth = i; } eh = std::max(eh, d); mh = std::max(mh, (double)fabsf(ha[i])); }
            if (nph == 2) for (int i = 0; i < inter; ++i) { const float x = h2f(aa[i]), y = h2f(ab[i]); const double d = fabsf(x - y); if (d > 4e-3 * fabsf(x) + 1e-3) { ++nbada; if (firsta < 0) firsta = i; } ea = std::max(ea, d); ma = std::max(ma, (double)fabsf(x)); }
            printf("chain of %d phase(s): h32 max|d| %.3e (max %.2f), %d rows off (first %d)", nph, eh, mh, nbadh, firsth);
            if (nph == 2) printf(" | act max|d| %.3e (max %.2f), %d outputs off (first %d)", ea, ma, nbada, firsta);
            printf("\n");
            if (nph == 2) {
                std::vector<float> sb(256); std::vector<unsigned long long> sg(256), xg(2048); std::vector<uint16_t> xb(hidden);
                CK(hipMemcpy(sb.data(), ssq, 256 * 4, hipMemcpyDeviceToHost));
                CK(hipMemcpy(sg.data(), gran[0] + gran_ssq_off, 256 * 8, hipMemcpyDeviceToHost));
                CK(hipMemcpy(xg.data(), gran[0], 2048 * 8, hipMemcpyDeviceToHost));
                CK(hipMemcpy(xb.data(), xn, hidden * 2, hipMemcpyDeviceToHost));
                double s1 = 0, s2 = 0; int tagbad = 0, xbad = 0;
                for (int i = 0; i < 256; ++i) { s1 += sb[i]; float f; uint32_t u = (uint32_t)sg[i]; memcpy(&f, &u, 4); s2 += f; if ((sg[i] >> 32) != (sg[0] >> 32)) ++tagbad; }
                for (int i = 0; i < 2048; ++i) { if ((uint16_t)xg[i] != xb[2 * i] || (uint16_t)(xg[i] >> 16) != xb[2 * i + 1]) ++xbad; if ((xg[i] >> 32) != (sg[0] >> 32)) ++tagbad; }
                {
                    ChArgs a2 = chain_args(w, h32b, qkvb, 2); a2.ph[1].y = act2; a2.dbg = dbg;
                    CK(hipMemcpy(h32b, h0, hidden * 4, hipMemcpyDeviceToDevice));
                    launch_chain<NW, D, true>(a2, smem, 0);
                    CK(hipDeviceSynchronize());
                    std::vector<long long> hd((size_t)nblk * CH_MAX_PHASES * 16);
                    CK(hipMemcpy(hd.data(), dbg, hd.size() * 8, hipMemcpyDeviceToHost));
                    double xs_ref = 0; for (int i = 0; i < hidden; ++i) xs_ref += h2f(xb[i]);
                    printf("   expected rs_norm %.7f, sum of xn %.4f; blocks 0, 1, 100, 255 saw:", 1.0 / sqrt(s1 / hidden + eps), xs_ref);
                    for (int b : {0, 1, 100, 255}) { uint32_t u1 = (uint32_t)hd[((size_t)b * CH_MAX_PHASES + 1) * 16 + 8], u2 = (uint32_t)hd[((size_t)b * CH_MAX_PHASES + 1) * 16 + 9]; float f1, f2; memcpy(&f1, &u1, 4); memcpy(&f2, &u2, 4); printf(" (%.7f, %.4f)", f1, f2); }
                    printf("\n");
                }
                printf("   sum of squares: launches %.6f, chain granules %.6f (tags differing %d); xn granules differing from the launches' xn: %d of 2048\n", s1, s2, tagbad, xbad);
                // mismatches by position of the row set inside its block (sets are dealt: first r blocks q + 1)
                const int nsets = 2 * inter / 16, q = nsets / nblk, r = nsets % nblk;
                int hist[8] = {0}, tot[8] = {0};
                for (int i = 0; i < inter; ++i) {
                    const int set = i / 8; int b, pos;
                    if (set < r * (q + 1)) { b = set / (q + 1); pos = set % (q + 1); } else { b = r + (set - r * (q + 1)) / q; pos = (set - r * (q + 1)) % q; }
                    (void)b;
                    const float x = h2f(aa[i]), y = h2f(ab[i]);
                    tot[pos]++; if (fabsf(x - y) > 4e-3 * fabsf(x) + 1e-3) hist[pos]++;
                }
                printf("   act mismatches by row-set position in the block:"); for (int k = 0; k < 6; ++k) printf(" [%d] %d/%d", k, hist[k], tot[k]); printf("\n   first outputs (launches vs chain):");
                for (int i = 0; i < 24; ++i) printf(" %.3f/%.3f", h2f(aa[i]), h2f(ab[i])); printf("\n");
            }
            if (nbadh && nbadh < 64) { printf("   rows off:"); for (int i = 0; i < hidden; ++i) if (fabsf(ha[i] - hb[i]) > 1e-3 * fabsf(ha[i]) + 1e-3) printf(" %d(%.3f vs %.3f)", i, ha[i], hb[i]); printf("\n"); }
        }
    }
    if (bad && !getenv("CHAIN_LAB_FORCE")) { printf("correctness failed: no timing\n"); return 2; }

    // ---- timing: interleaved rounds
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time_us = [&](auto f, int reps) {
        for (int l = 0; l < L; ++l) f(l);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < reps; ++r) for (int l = 0; l < L; ++l) f(l);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        return ms * 1e3f / (reps * L);
    };
    std::vector<float> tb, tc, tc3, tb3, tcn;
    for (int round = 0; round < 7; ++round) {
        tb.push_back(time_us([&](int l) { baseline(W[l], h32a, qkva, 0); }, 10));
        tc.push_back(time_us([&](int l) { launch_chain<NW, D, false>(chain_args(W[l], h32b, qkvb, 4), smem, 0); }, 10));
        tc3.push_back(time_us([&](int l) { launch_chain<NW, D, false>(chain_args(W[l], h32b, qkvb, 2), smem, 0); }, 10));
        tcn.push_back(time_us([&](int l) { launch_chain<NW, D, false, true>(chain_args(W[l], h32b, qkvb, 4), smem, 0); }, 10));
        tb3.push_back(time_us([&](int l) {
            const Layer& w = W[l];
            QK(qeft_decode_linear(att, w.o.qw, w.o.szp, w.o.ow, nullptr, h32a, hidden, hidden, 128, 128, 0, h32a, nullptr, 0, eps, gam1, xn, ssq, 0));
            QK(qeft_decode_linear(xn, w.gu.qw, w.gu.szp, w.gu.ow, nullptr, act, 2 * inter, hidden, 128, 128, 1, nullptr, ssq, n_ssq, eps, nullptr, nullptr, nullptr, 0));
        }, 10));
    }
    auto med = [](std::vector<float> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    auto mn = [](std::vector<float> v) { return *std::min_element(v.begin(), v.end()); };
    const double bytes4 = 9715712.0 + 2 * 26097152.0 + 24770048.0 + (9715712.0 * 3 - 2 * 2 * 4096 * 0);   // o + gate|up + down + q|k|v (3 x 4096^2)
    printf("four launches (o, gate|up, down, q|k|v): median %.2f us  min %.2f us   -> %.0f GB/s\n", med(tb), mn(tb), bytes4 / med(tb) / 1e3);
    printf("one chain launch (4 phases)            : median %.2f us  min %.2f us   -> %.0f GB/s\n", med(tc), mn(tc), bytes4 / med(tc) / 1e3);
    printf("the same chain with the MATH REMOVED (loads, ring, edges, barriers kept; garbage results): median %.2f us  min %.2f us\n", med(tcn), mn(tcn));
    printf("edge pair o_proj -> gate|up: two launches median %.2f us, chain of 2 phases median %.2f us\n", med(tb3), med(tc3));
    uint32_t stv = 0; CK(hipMemcpy(&stv, status, 4, hipMemcpyDeviceToHost));
    printf("status after timing: %#x\n", stv);

    // ---- timeline of the chain (stamps of every block, 100 MHz)
    {
        std::vector<long long> h((size_t)nblk * CH_MAX_PHASES * 16);
        for (int rep = 0; rep < 3; ++rep)
            for (int l = 0; l < 3; ++l) {
                ChArgs a = chain_args(W[l], h32b, qkvb, 4);
                a.dbg = dbg;
                launch_chain<NW, D, true>(a, smem, 0);
            }
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
        long long t0 = 1LL << 62;
        for (int b = 0; b < nblk; ++b) t0 = std::min(t0, h[((size_t)b * CH_MAX_PHASES + 0) * 16 + 0]);
        const char* names[4] = {"o_proj", "gate|up", "down", "q|k|v"};
        printf("timeline of one chain launch (us after the first block's entry; min..max over the %d blocks):\n", nblk);
        for (int p = 0; p < 4; ++p) {
            long long mnv[8], mxv[8];
            for (int i = 0; i < 8; ++i) { mnv[i] = 1LL << 62; mxv[i] = 0; }
            for (int b = 0; b < nblk; ++b)
                for (int i = 0; i < 7; ++i) { const long long v = h[((size_t)b * CH_MAX_PHASES + p) * 16 + i]; mnv[i] = std::min(mnv[i], v); mxv[i] = std::max(mxv[i], v); }
            auto us = [&](long long v) { return (v - t0) / 100.0; };
            printf("  %-8s wave 0 ready to sweep %.2f..%.2f | sweep done %.2f..%.2f | past B1 %.2f..%.2f | steps done %.2f..%.2f | past B2 %.2f..%.2f | published %.2f..%.2f\n",
                   names[p], us(mnv[1]), us(mxv[1]), us(mnv[2]), us(mxv[2]), us(mnv[3]), us(mxv[3]), us(mnv[4]), us(mxv[4]),
                   us(mnv[5]), us(mxv[5]), us(mnv[6]), us(mxv[6]));
        }
    }
    return bad ? 2 : 0;
}
