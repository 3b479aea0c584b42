// mcq_api.hip -- C ABI (include/mcq.h) over the gfx950 kernels of mcq_kernels.h.
// Host side only enqueues kernels on the caller's stream; see mcq.h for the contract.
#include "../../include/mcq.h"
#include "mcq_kernels.h"
#include "mcq_fix_kernels.h"
#include "mcq_loss_kernels.h"
#include "mcq_tf_kernels.h"
#include "mcq_pass16_kernels.h"
#include "mcq_train_kernels.h"

#include <cstdlib>
#include <vector>

using namespace mcq;

namespace {

inline int round_up16(int d) { return (d + 15) & ~15; }
inline size_t align256(size_t v) { return (v + 255) & ~size_t(255); }
inline bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

// K_cutoff rule of _refine_indexes (quantization/quantization.py:453-463)
int k_cutoff(int K, int L) {
    int kc = (K <= 16) ? 8 : 16;
    while (L >= 4) { L /= 4; kc *= 2; }
    return kc < 128 ? kc : 128;
}

struct Prepared {
    const float *C, *Q, *bias, *scales, *G, *mean, *wmu;
    const int8_t *Cf, *Wf;      // limb planes of the scaled centers / of to_logits.weight (mcq_fix_kernels.h)
    const int *Ce, *We;         // their row exponents
};

struct PreparedLayout {
    size_t offC, offQ, offCf, offCe, offWf, offWe, offWmu, offBias, offScales, offMean, offCMean, offG, total;
};

PreparedLayout prepared_layout(int N, int K, int D) {
    const size_t nk = (size_t)N * K, Dp = round_up16(D);
    const size_t planes = fix_plane_bytes((long)nk, D), exps = (size_t)fix_round_rows((long)nk) * 4;
    PreparedLayout l;
    l.offC = 0;
    l.offQ = align256(l.offC + nk * Dp * 4);
    l.offCf = align256(l.offQ + nk * 4);
    l.offCe = align256(l.offCf + planes);
    l.offWf = align256(l.offCe + exps);
    l.offWe = align256(l.offWf + planes);
    l.offWmu = align256(l.offWe + exps);          // fixdot(data mean, W[r]), float[nk] (the logits product reads centered frames)
    l.offBias = align256(l.offWmu + nk * 4);
    l.offScales = align256(l.offBias + nk * 4);   // float[2] {cscale_exp, lscale_exp} (mcq_prepare_dev)
    l.offMean = align256(l.offScales + 8);        // get_data_mean() of the scaled centers, float[Dp]
    l.offCMean = align256(l.offMean + Dp * 4);    // the codebooks' own means mu_n, float[N][Dp] (mean = mu_0 + mu_1 + ...)
    l.offG = align256(l.offCMean + (size_t)N * Dp * 4);   // Gram matrix G[nk][nk] of the CENTERED rows C[n][k] - mu_n
    l.total = align256(l.offG + nk * nk * 4);
    // k_fgemm's epilogue DMA-copies 128 bias floats per row tile without a bound check: rows past N*K of the last tile read what
    // FOLLOWS the bias (scale factors, means, the Gram matrix: finite floats inside this blob; such rows belong to no codebook and
    // their values are never stored).  The layout guarantees those bytes exist:
    static_assert(kFixTile == 128, "");
    if (l.total < l.offBias + (nk + kFixTile) * 4) l.total = align256(l.offBias + (nk + kFixTile) * 4);
    return l;
}

Prepared prepared_view(const void *p, int N, int K, int D) {
    const PreparedLayout l = prepared_layout(N, K, D);
    const char *b = static_cast<const char *>(p);
    return Prepared{reinterpret_cast<const float *>(b + l.offC), reinterpret_cast<const float *>(b + l.offQ),
                    reinterpret_cast<const float *>(b + l.offBias), reinterpret_cast<const float *>(b + l.offScales),
                    reinterpret_cast<const float *>(b + l.offG), reinterpret_cast<const float *>(b + l.offMean),
                    reinterpret_cast<const float *>(b + l.offWmu),
                    reinterpret_cast<const int8_t *>(b + l.offCf), reinterpret_cast<const int8_t *>(b + l.offWf),
                    reinterpret_cast<const int *>(b + l.offCe), reinterpret_cast<const int *>(b + l.offWe)};
}

// CT: how a codebook entry is held (mcq_tf_kernels.h: one byte up to 256 entries per codebook, two above)
template <typename CT>
struct WorkspaceT {
    CT *idx, *idxB, *idxC, *final_idx;        // B, C, final: fixed-point skipping only
    int *map[2], *cnt;
    float *E, *R, *xx, *XC;                   // per vector: |x_err|^2, |x_err - old_n|^2, |x|^2, x.C products
    float *gterms;                            // per vector: the N*N Gram entries G[o_m][o_m2] of the current indexes
    int8_t *xf;                               // limb planes of the CENTERED frames of a chunk (x - mean: both products read them)
    int *xe;                                  // and their row exponents
    float *tabs[2];                           // group tables of two consecutive levels (ping-pong)
    TfLists tf;                               // candidate lists of every level
};

int tf_levels(int N) { int v = 0; while ((1 << v) < N) ++v; return v; }   // lists exist at levels 0 .. tf_levels(N) - 1

// floats of the largest set of group tables any combine needs at one level (per vector)
size_t tf_tab_floats(int N, int K) {
    const int nlev = tf_levels(N);
    size_t best = 0;
    for (int v = 2; v < nlev; ++v) {
        const size_t groups = (size_t)N >> (v + 1);
        for (int u = 1; u < v; ++u) {
            const size_t per = (size_t)1 << (v - u), kc = k_cutoff(K, 1 << u);
            const size_t f = groups * per * per * kc * kc;
            best = f > best ? f : best;
        }
    }
    return best;
}

inline size_t code_bytes_of(int K) { return K > 256 ? 2 : 1; }

size_t workspace_per_vector(int N, int K, int D) {
    // idx x4, maps, E, xx, R, XC, lists (entries / positions / scores: <= 16 cb + 2*16 + 4*16 bytes per codebook and level), tabs x2,
    // the frame as limb planes + its exponent, the N*N Gram terms of E / R
    const size_t cb = code_bytes_of(K);
    return 4 * cb * (size_t)N + 8 + 8 + 4 * (size_t)N + 4 * (size_t)N * K + (size_t)tf_levels(N) * N * (16 * cb + 2 * 16 + 4 * 16) + 64 +
           2 * 4 * tf_tab_floats(N, K) + 4 * (size_t)fix_round_cols(D) + 4 + 4 * (size_t)N * N;
}
// alignment of the carved arrays + the rows the limb planes are padded by (to a multiple of 128)
size_t workspace_slack(int D) { return 48 * 256 + (size_t)kFixTile * (4 * (size_t)fix_round_cols(D) + 4); }

// default chunk: 65,536 vectors, fewer when a vector's share of the workspace is large (N >= 32), so that the workspace
// mcq_encode_workspace_bytes asks for stays near 2 GB
long default_chunk(int N, int K, int D) {
    long c = (long)(((size_t)2 << 30) / workspace_per_vector(N, K, D));
    c = c > 65536 ? 65536 : c;
    c = c < 1024 ? 1024 : c;
    return c & ~127L;
}

template <typename CT>
WorkspaceT<CT> carve(void *ws, long Bc, int N, int K, int D) {
    char *p = static_cast<char *>(ws);
    size_t off = 0;
    auto take = [&](size_t bytes) { char *q = p + off; off = align256(off + bytes); return q; };
    WorkspaceT<CT> w;
    w.idx = reinterpret_cast<CT *>(take((size_t)Bc * N * sizeof(CT)));
    w.idxB = reinterpret_cast<CT *>(take((size_t)Bc * N * sizeof(CT)));
    w.idxC = reinterpret_cast<CT *>(take((size_t)Bc * N * sizeof(CT)));
    w.final_idx = reinterpret_cast<CT *>(take((size_t)Bc * N * sizeof(CT)));
    for (int i = 0; i < 2; ++i) w.map[i] = reinterpret_cast<int *>(take((size_t)Bc * 4));
    w.cnt = reinterpret_cast<int *>(take(64 * 4));
    w.E = reinterpret_cast<float *>(take((size_t)Bc * 4));
    w.R = reinterpret_cast<float *>(take((size_t)Bc * N * 4));
    w.xx = reinterpret_cast<float *>(take((size_t)Bc * 4));
    w.gterms = reinterpret_cast<float *>(take((size_t)Bc * N * N * 4));
    w.XC = reinterpret_cast<float *>(take((size_t)Bc * N * K * 4));
    w.xf = reinterpret_cast<int8_t *>(take(fix_plane_bytes(Bc, D)));
    w.xe = reinterpret_cast<int *>(take((size_t)fix_round_rows(Bc) * 4));
    const int nlev = tf_levels(N);
    w.tf.ent = nullptr;
    w.tf.out_i64 = nullptr;
    w.tf.out_u8 = nullptr;
    w.tf.erG = w.tf.erXC = w.tf.erxx = nullptr;
    w.tf.erE = w.tf.erR = nullptr;
    w.tf.erK = 0;
    for (int v = 0; v < kTfLevels; ++v) {
        w.tf.kc[v] = k_cutoff(K, 1 << v);
        w.tf.pos[v] = nullptr;
        w.tf.S[v] = nullptr;
        if (v >= nlev) continue;                       // no list of that level
        const int kc = w.tf.kc[v];
        if (v == 0) w.tf.ent = reinterpret_cast<uint8_t *>(take((size_t)Bc * N * kc * sizeof(CT)));
        else w.tf.pos[v] = reinterpret_cast<uint8_t *>(take((size_t)Bc * (N >> v) * kc * 2));
        w.tf.S[v] = reinterpret_cast<float *>(take((size_t)Bc * (N >> v) * kc * 4));
    }
    const size_t tf = tf_tab_floats(N, K);
    for (int i = 0; i < 2; ++i) w.tabs[i] = tf ? reinterpret_cast<float *>(take((size_t)Bc * tf * 4)) : nullptr;
    return w;
}

// up to 64 codebooks (QuantizerTrainer produces at most 64 x 16 and 32 x 256: bytes_per_frame <= 32); codebooks of 512 and 1,024
// entries (Quantizer(codebook_size = ...) used with as_bytes = False) as long as the Gram matrix stays within 16,384 rows (1 GB)
bool domain_ok(int N, int K, int D) {
    return is_pow2(K) && K >= 16 && K <= 1024 && is_pow2(N) && N <= 64 && (long)N * K <= 16384 && D >= 1 && D <= 16384;
}
int domain_err(int N, int K, int D = 1) {
    return (K < 16 || K > 1024 || N > 64 || (long)N * K > 16384 || D > 16384) ? MCQ_EUNSUPPORTED : MCQ_EINVAL;
}

// optional per-launch timing (mcq_profile_encode)
struct Prof {
    hipStream_t stream;
    int only = -1;                            // the one category this encode times (-1: all)
    std::vector<hipEvent_t> ev;
    std::vector<int> cat, first, last;        // interval i: events first[i] .. last[i]
    int open = -1;
    // An event pair round EVERY launch keeps the launches from overlapping their predecessor's tail (the profiled encode took
    // 9-11 % longer than the timed one), so mcq_profile_encode runs one encode per category and times only that category's
    // launches in it; two timed launches that follow each other share the event between them.
    void begin(int category) {
        if (only >= 0 && category != only) { open = -1; return; }      // (a launch that is not timed separates two that are)
        if (ev.empty() || open != (int)ev.size() - 1) {
            hipEvent_t e;
            (void)hipEventCreate(&e);
            (void)hipEventRecord(e, stream);
            ev.push_back(e);
        }
        open = (int)ev.size() - 1;
    }
    void untimed() { open = -1; }             // something was enqueued outside every category: the next interval takes a fresh event
    void end(int category) {
        if (only >= 0 && category != only) { open = -1; return; }
        hipEvent_t e;
        (void)hipEventCreate(&e);
        (void)hipEventRecord(e, stream);
        ev.push_back(e);
        first.push_back(open);
        last.push_back((int)ev.size() - 1);
        cat.push_back(category);
        open = (int)ev.size() - 1;
    }
};

thread_local int g_last_launches = 0;

#define MCQ_LAUNCH_CHECK()                               \
    do {                                                 \
        hipError_t e_ = hipGetLastError();               \
        if (e_ != hipSuccess) return (int)e_;            \
        ++g_last_launches;                               \
    } while (0)

// rows -> limb planes + exponents (+ |row|^2): the operands of every product of the path
FixRowsArgs fix_rows_args(const float *src, int xh, long R, int D, long ld, int8_t *planes, int *exps, float *xx,
                          const float *bias_src = nullptr, float *bias_dst = nullptr, const float *sub = nullptr,
                          long sub_per = 0, long sub_ld = 0, const float *dot_vec = nullptr, float *dot_out = nullptr) {
    return FixRowsArgs{src, xh, R, fix_round_rows(R), D, ld, fix_round_cols(D), planes, exps, xx, bias_src, bias_dst,
                       sub, sub_per, sub_ld, dot_vec, dot_out};
}

// sub != nullptr: the rows are centered by sub[0 .. D) (the frames of the search and of the logits: x - mean)
int launch_fix_rows(const float *src, int xh, long R, int D, long ld, int8_t *planes, int *exps, float *xx, hipStream_t st,
                    const float *sub = nullptr) {
    const FixRowsArgs a = fix_rows_args(src, xh, R, D, ld, planes, exps, xx, nullptr, nullptr, sub, 0, 0);
    // four rows per workgroup, one per wave (16 rows per workgroup and 256-byte runs into the planes measured slower:
    // 0.086 vs 0.071 ms at 65,536 x 512)
    hipLaunchKernelGGL(k_fix_rows<4>, dim3((unsigned)(a.Rp / 4)), dim3(256), 0, st, a);
    MCQ_LAUNCH_CHECK();
    return 0;
}

int launch_fix_rows2(const FixRowsArgs &a, const FixRowsArgs &b, hipStream_t st) {
    const unsigned na = (unsigned)(a.Rp / 4), nb = (unsigned)(b.Rp / 4);
    hipLaunchKernelGGL(k_fix_rows2<4>, dim3(na + nb), dim3(256), 0, st, a, b, na);
    MCQ_LAUNCH_CHECK();
    return 0;
}

// the fixed-point GEMM: persistent workgroups of eight waves, one per CU
template <int MODE>
int launch_fgemm(FixGemm g, hipStream_t st) {
    // the kernel's 132 KB of dynamic LDS has to be allowed once per device (a process may drive several)
    static bool allowed[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 63;
    if (!allowed[dev] || dev == 63) {
        const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fgemm<MODE>),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, kFixLds);
        if (attr != hipSuccess) return (int)attr;
        allowed[dev] = true;
    }
    const long MT = g.RA / kFixTile, NT = g.RB / kFixTile;
    const int H = (MODE == FG_LOGITS && g.K > kFixTile) ? g.K / kFixTile : 1;
    const long big = g.walk_rows ? NT : MT, small_units = (g.walk_rows ? MT : NT) / H;
    long units = (big + 7) / 8 * small_units;          // per XCD
    if (units > 32) units = 32;                        // 32 CUs per XCD, one workgroup each
    hipLaunchKernelGGL((k_fgemm<MODE>), dim3((unsigned)(8 * units)), dim3(512), kFixLds, st, g);
    MCQ_LAUNCH_CHECK();
    return 0;
}

// XC[b][r] = fixdot(x_b, C_r) (or the Gram matrix with the centers as "frames"): frames stream, the centers are the table.
// (The other arrangement -- centers as the tile rows, four consecutive centers of a frame as one 16-byte store, as the
// logits are laid out -- measured 0.600 against 0.581 ms with eight waves; with four it was the faster one, 0.712 / 0.746.)
int launch_xc(const int8_t *xf, const int *xe, long B, const int8_t *Cf, const int *Ce, long nk, int D, float *out,
              hipStream_t st) {
    FixGemm g{};
    g.A = xf; g.ea = xe; g.RA = fix_round_rows(B); g.M = B;
    g.B = Cf; g.eb = Ce; g.RB = fix_round_rows(nk); g.N = nk;
    g.Dq = fix_round_cols(D);
    g.walk_rows = 0;
    g.out = out; g.ldo = nk;
    return launch_fgemm<FG_STORE>(g, st);
}

// logits[b][r] = fixdot(x_b, W_r) * lscale + bias[r] (stored when logits != nullptr) and the arg max per codebook
int launch_logits(const int8_t *xf, const int *xe, long B, const Prepared &P, int N, int K, int D, float lscale,
                  const float *lscale_ptr, float *logits, void *idx, hipStream_t st) {
    const long nk = (long)N * K;
    FixGemm g{};
    g.A = P.Wf; g.ea = P.We; g.RA = fix_round_rows(nk); g.M = nk;
    g.B = xf; g.eb = xe; g.RB = fix_round_rows(B); g.N = B;
    g.Dq = fix_round_cols(D);
    g.walk_rows = 1;
    g.bias = P.bias; g.wmu = P.wmu; g.lscale = lscale; g.lscale_ptr = lscale_ptr;
    g.logits = logits; g.ldo = nk; g.idx = idx; g.idx_wide = K > 256 ? 1 : 0; g.K = K; g.ncb = N;
    return launch_fgemm<FG_LOGITS>(g, st);
}

// ---------------------------------------------------------------- the refinement pass
template <int K>
int launch_tf_stage0_k(int N, const float *G, const float *XC, const tf_code_of<K> *idx, const float *R, const float *Q, long B,
                       int keep, tf_code_of<K> *ent, float *S, tf_code_of<K> *fin, const int *nact, const int *map, hipStream_t st) {
    const dim3 grid((unsigned)(((B + 3) / 4) * N)), block(256);
#define MCQ_S0_CASE(NN) \
    case NN: hipLaunchKernelGGL((k_tf_stage0<K, NN>), grid, block, 0, st, G, XC, idx, R, Q, B, keep, ent, S, fin, nact, map); break;
    switch (N) {
        MCQ_S0_CASE(1) MCQ_S0_CASE(2) MCQ_S0_CASE(4) MCQ_S0_CASE(8) MCQ_S0_CASE(16) MCQ_S0_CASE(32) MCQ_S0_CASE(64)
        default: return MCQ_EUNSUPPORTED;
    }
#undef MCQ_S0_CASE
    MCQ_LAUNCH_CHECK();
    return 0;
}

int launch_tf_stage0(int K, int N, const float *G, const float *XC, const uint16_t *idx, const float *R, const float *Q,
                     long B, int keep, uint16_t *ent, float *S, uint16_t *fin, const int *nact, const int *map, hipStream_t st) {
    switch (K) {
        case 512: return launch_tf_stage0_k<512>(N, G, XC, idx, R, Q, B, keep, ent, S, fin, nact, map, st);
        case 1024: return launch_tf_stage0_k<1024>(N, G, XC, idx, R, Q, B, keep, ent, S, fin, nact, map, st);
        default: return MCQ_EUNSUPPORTED;
    }
}

int launch_tf_stage0(int K, int N, const float *G, const float *XC, const uint8_t *idx, const float *R, const float *Q,
                     long B, int keep, uint8_t *ent, float *S, uint8_t *fin, const int *nact, const int *map, hipStream_t st) {
    if (K == 16 && N >= 4 && keep <= 16) {       // four codebooks per wave, rank-in-row selection
        const dim3 grid((unsigned)((B * (N / 4) + 3) / 4)), block(256);
#define MCQ_S16_CASE(NN) \
    case NN: hipLaunchKernelGGL((k_tf_stage0_k16<NN>), grid, block, 0, st, G, XC, idx, R, Q, B, keep, ent, S, nact, map); break;
        switch (N) {
            MCQ_S16_CASE(4) MCQ_S16_CASE(8) MCQ_S16_CASE(16) MCQ_S16_CASE(32) MCQ_S16_CASE(64)
            default: return MCQ_EUNSUPPORTED;
        }
#undef MCQ_S16_CASE
        MCQ_LAUNCH_CHECK();
        return 0;
    }
    switch (K) {
        case 16: return launch_tf_stage0_k<16>(N, G, XC, idx, R, Q, B, keep, ent, S, fin, nact, map, st);
        case 32: return launch_tf_stage0_k<32>(N, G, XC, idx, R, Q, B, keep, ent, S, fin, nact, map, st);
        case 64: return launch_tf_stage0_k<64>(N, G, XC, idx, R, Q, B, keep, ent, S, fin, nact, map, st);
        case 128: return launch_tf_stage0_k<128>(N, G, XC, idx, R, Q, B, keep, ent, S, fin, nact, map, st);
        case 256: return launch_tf_stage0_k<256>(N, G, XC, idx, R, Q, B, keep, ent, S, fin, nact, map, st);
        default: return MCQ_EUNSUPPORTED;
    }
}

template <typename CT>
int launch_tf_er(int N, const float *G, const float *XC, const CT *idx, const float *xx, long B, int K, float *E, float *R,
                 float *gterms, const int *nact, const int *map, hipStream_t st) {
    const dim3 grid((unsigned)((B + 3) / 4)), block(256);
    const bool direct = B <= 8192;          // one launch instead of two (E / R of a trainer batch: 4.9 + 4.7 -> about 5 us)
#define MCQ_ER_CASE(NN)                                                                                                  \
    case NN:                                                                                                             \
        if (!direct) {                                                                                                   \
            hipLaunchKernelGGL((k_tf_gram_terms<NN, CT>), dim3((unsigned)(((B + 4 * (64 / NN) - 1) / (4 * (64 / NN))) * NN)), block, 0, st, G, idx, \
                               B, K, gterms, nact);                                                                      \
            MCQ_LAUNCH_CHECK();                                                                                          \
        }                                                                                                                \
        hipLaunchKernelGGL((k_tf_er<NN, CT>), grid, block, 0, st, gterms, XC, idx, xx, B, K, E, R, nact, map,              \
                           direct ? G : static_cast<const float *>(nullptr));                                           \
        break;
    switch (N) {
        MCQ_ER_CASE(1) MCQ_ER_CASE(2) MCQ_ER_CASE(4) MCQ_ER_CASE(8) MCQ_ER_CASE(16) MCQ_ER_CASE(32) MCQ_ER_CASE(64)
        default: return MCQ_EUNSUPPORTED;
    }
#undef MCQ_ER_CASE
    MCQ_LAUNCH_CHECK();
    return 0;
}

// group tables of level u >= 2 from those of level u - 1 (list lengths kh -> kc)
int launch_tf_up(int kh, int kc, const TfLists &L, long B, int N, int u, int ntab, int per, const float *in, float *out,
                 const int *nact, hipStream_t st) {
    const dim3 grid((unsigned)(B * ntab)), block(64);
#define MCQ_UP_CASE(A, C) \
    if (kh == A && kc == C) { hipLaunchKernelGGL((k_tf_up<A, C>), grid, block, 0, st, L, B, N, u, ntab, per, in, out, nact); MCQ_LAUNCH_CHECK(); return 0; }
    MCQ_UP_CASE(16, 32) MCQ_UP_CASE(32, 32) MCQ_UP_CASE(32, 64) MCQ_UP_CASE(8, 16) MCQ_UP_CASE(16, 16)
#undef MCQ_UP_CASE
    return MCQ_EUNSUPPORTED;
}

// combine of the siblings of level v >= 2 (list lengths kh at level v - 1, kc at level v)
template <typename CT>
int launch_tf_comb(int kh, int kc, const float *E, const TfLists &L, long B, int N, int v, int keep, const float *tabs,
                   CT *fin, const int *nact, hipStream_t st) {
    const dim3 grid((unsigned)(B * (N >> (v + 1)))), block(64);
#define MCQ_COMB_CASE(A, C)                                                                                                        \
    if (kh == A && kc == C) {                                                                                                     \
        if (fin) hipLaunchKernelGGL((k_tf_comb<A, C, true, CT>), grid, block, 0, st, E, L, B, N, v, keep, tabs, fin, nact);       \
        else hipLaunchKernelGGL((k_tf_comb<A, C, false, CT>), grid, block, 0, st, E, L, B, N, v, keep, tabs, fin, nact);          \
        MCQ_LAUNCH_CHECK();                                                                                                       \
        return 0;                                                                                                                 \
    }
    MCQ_COMB_CASE(16, 32) MCQ_COMB_CASE(32, 32) MCQ_COMB_CASE(32, 64) MCQ_COMB_CASE(64, 64)
    if constexpr (sizeof(CT) == 1) { MCQ_COMB_CASE(8, 16) MCQ_COMB_CASE(16, 16) }      // (lists of 8: 16-entry codebooks)
#undef MCQ_COMB_CASE
    return MCQ_EUNSUPPORTED;
}

// profiling categories (mcq_profile_encode): one per KIND OF LAUNCH of the shipped sequence -- the profiler records events round
// the launches an encode makes and changes none of them
enum { CAT_LOGITS = 0, CAT_XX = 1, CAT_STAGE0 = 2, CAT_XC = 3, CAT_LEVEL0 = 4, CAT_LEVEL1 = 5, CAT_TABLES = 6, CAT_COMBINE = 7,
       CAT_TABLES_UP = 8, CAT_COMBINE_UP = 9, CAT_ER = 10, CAT_LEVEL1_FUSED = 11, CAT_TAIL = 12, CAT_COUNT = 13 };
const char *const kCatNames[CAT_COUNT] = {
    "logits_product_argmax",      // k_fgemm<FG_LOGITS>
    "frames_to_limbs",            // k_fix_rows
    "stage0_tables",              // k_tf_stage0 / k_tf_stage0_k16
    "xc_product",                 // k_fgemm<FG_STORE>
    "combine_level0",             // k_tf_pair0
    "combine_level1",             // k_tf_pair1 (4 codebooks: the last combine)
    "tables_level1",              // k_tf_table1 (more than 16 codebooks)
    "combine_level2",             // k_tf_comb at level 2 (8 codebooks: the last combine, which also forms E / R of the next pass)
    "tables_upper_levels",        // k_tf_table1 / k_tf_up above level 2
    "combine_upper_levels",       // k_tf_comb / k_tf_comb3 above level 2
    "residual_energies",          // k_tf_gram_terms + k_tf_er (first pass of a call; later passes: inside the last combine)
    "level1_combines_and_tables", // k_tf_level1: the level-1 combines and the cousin tables of level 2 in one launch
    "encode_tail",                // k_finalize / k_import_indexes / k_compact
};

// the combines of one refinement pass; lists of K >= 32 hold 16, 16, 32, 32, 64 candidates, of K == 16: 8, 8, 16, 16, 32, 32
// (the kernels over lists of 8 -- 16-entry codebooks -- exist for one-byte entries only)
#define MCQ_TF_LAUNCH2(SMALLK, BIGK, grid, block, ...)                                                       \
    do {                                                                                                     \
        bool s_ = false;                                                                                     \
        if constexpr (sizeof(CT) == 1) {                                                                     \
            if (small) { hipLaunchKernelGGL(SMALLK, grid, block, 0, st, __VA_ARGS__); s_ = true; }           \
        }                                                                                                    \
        if (!s_) hipLaunchKernelGGL(BIGK, grid, block, 0, st, __VA_ARGS__);                                  \
    } while (0)
// MCQ_PASS16=0: the separate kernels for 16 x 16 codebooks as well (same-box A/B of the LDS-resident pass; identical results)
inline bool pass16_enabled() {
    static const bool on = !(getenv("MCQ_PASS16") && atoi(getenv("MCQ_PASS16")) == 0);
    return on;
}

template <typename CT>
int run_tf_combines(const float *G, const CT *idx_cur, CT *idx_new, const WorkspaceT<CT> &w, const TfLists &L, long B, int N,
                    int K, const int *nact, hipStream_t st, Prof *prof) {
    const bool small = (K == 16);
    const int nlev = tf_levels(N);
    {   // level 0: single codebooks
        const int keep = (N == 2) ? 1 : L.kc[1];
        CT *fin = (N == 2) ? idx_new : nullptr;
        const dim3 grid((unsigned)(B * (N / 2)));
        if (prof) prof->begin(CAT_LEVEL0);
        bool done0 = false;
        if constexpr (sizeof(CT) == 1) {
            // lists of 16 one-byte entries: the slot-major kernel (a 16-lane group of a gather = one table row: a fifth faster than
            // the lane-major one, profiles/r06_ab_pair0_slot_major.txt)
            if (!small) {
                hipLaunchKernelGGL(k_tf_pair0s, grid, dim3(64), 0, st, G, idx_cur, w.E, L, B, N, K, keep, fin, nact);
                done0 = true;
            }
        }
        if (!done0) MCQ_TF_LAUNCH2((k_tf_pair0<8, CT>), (k_tf_pair0<16, CT>), grid, dim3(64), G, idx_cur, w.E, L, B, N, K, keep, fin, nact);
        MCQ_LAUNCH_CHECK();
        if (prof) prof->end(CAT_LEVEL0);
    }
    // the level-1 combines and the cousin tables of level 2 share a launch (k_tf_level1): same workgroup-to-XCD mapping as the two
    // launches, one boundary and one tail less (5.95 -> 5.86 ms per encode of 65,536 vectors, 0.55 -> 0.51 ms at 4,096)
    const bool fuse_l1 = (N >= 8);
    // 16 codebooks: the 16 cousin tables of level 3 ride along (into tabs[1]).  With 256-entry codebooks at 65,536 vectors
    // that launch is 0.93 ms long and the merge bought nothing (23.99 / 23.92 against 23.90 / 24.0 ms per encode); a trainer
    // step of the first phase (16 x 16 codebooks, 4,096 vectors) saves a 29 us launch per pass
    const bool fuse_l3 = fuse_l1 && N == 16 && small;
    if (fuse_l1) {
        const int keep = L.kc[2];
        const int groups2 = N >> 3, per1 = 2, ntab1 = groups2 * per1 * per1;
        const unsigned pair_blocks = (unsigned)(B * (N / 4)), tab_blocks = (unsigned)(B * ntab1);
        const int ntab3 = fuse_l3 ? 16 : 1, per3 = fuse_l3 ? 4 : 1;
        const dim3 grid(pair_blocks + tab_blocks + (fuse_l3 ? (unsigned)(B * ntab3) : 0u));
        if (prof) prof->begin(CAT_LEVEL1_FUSED);
        MCQ_TF_LAUNCH2((k_tf_level1<8, 8, CT>), (k_tf_level1<16, 16, CT>), grid, dim3(64), G, idx_cur, w.E, L, B, N, K, keep, ntab1, per1, w.tabs[0],
                       nact, pair_blocks, tab_blocks, ntab3, per3, w.tabs[1]);
        MCQ_LAUNCH_CHECK();
        if (prof) prof->end(CAT_LEVEL1_FUSED);
    } else if (N >= 4) {   // level 1: pairs of codebooks
        const int keep = (N == 4) ? 1 : L.kc[2];
        CT *fin = (N == 4) ? idx_new : nullptr;
        const dim3 grid((unsigned)(B * (N / 4)));
        if (prof) prof->begin(CAT_LEVEL1);
        MCQ_TF_LAUNCH2((k_tf_pair1<8, 8, CT>), (k_tf_pair1<16, 16, CT>), grid, dim3(64), G, idx_cur, w.E, L, B, N, K, keep, fin, nact);
        MCQ_LAUNCH_CHECK();
        if (prof) prof->end(CAT_LEVEL1);
    }
    for (int v = 2; v < nlev; ++v) {   // level v: level-1 tables of the cousins below, raised level by level, then the combine
        const int groups = N >> (v + 1);
        const bool last = (v == nlev - 1);
        const int keep = last ? 1 : L.kc[v + 1];
        CT *fin = last ? idx_new : nullptr;
        const int per1 = 1 << (v - 1), ntab1 = groups * per1 * per1;
        const int cat_tab = (v == 2) ? CAT_TABLES : CAT_TABLES_UP, cat_comb = (v == 2) ? CAT_COMBINE : CAT_COMBINE_UP;
        if (!(fuse_l1 && v == 2) && !(fuse_l3 && v == 3)) {     // (these tables came with the level-1 combines)
            if (prof) prof->begin(cat_tab);
            MCQ_TF_LAUNCH2((k_tf_table1<8, 8, CT>), (k_tf_table1<16, 16, CT>), dim3((unsigned)(B * ntab1)), dim3(64), G, idx_cur, L, B, N, K, ntab1, per1,
                           w.tabs[0], nact);
            MCQ_LAUNCH_CHECK();
            if (prof) prof->end(cat_tab);
        }
        if (N == 16 && v == 3) {       // two groups of eight: levels 2 and 3 in one kernel, tables in LDS
            if (prof) prof->begin(cat_comb);
            const float *t3 = fuse_l3 ? w.tabs[1] : w.tabs[0];
            MCQ_TF_LAUNCH2((k_tf_comb3<8, 16, 16, CT>), (k_tf_comb3<16, 32, 32, CT>), dim3((unsigned)B), dim3(256), idx_cur, w.E, L, B, N, t3, idx_new, nact);
            MCQ_LAUNCH_CHECK();
            if (prof) prof->end(cat_comb);
            continue;
        }
        int cur = 0;
        for (int u = 2; u < v; ++u) {
            const int per = 1 << (v - u), ntab = groups * per * per;
            if (prof) prof->begin(cat_tab);
            const int rc = launch_tf_up(L.kc[u - 1], L.kc[u], L, B, N, u, ntab, per, w.tabs[cur], w.tabs[cur ^ 1], nact, st);
            if (rc) return rc;
            if (prof) prof->end(cat_tab);
            cur ^= 1;
        }
        if (prof) prof->begin(cat_comb);
        const int rc = launch_tf_comb<CT>(L.kc[v - 1], L.kc[v], w.E, L, B, N, v, keep, w.tabs[cur], fin, nact, st);
        if (rc) return rc;
        if (prof) prof->end(cat_comb);
    }
    return 0;
}
#undef MCQ_TF_LAUNCH2

template <typename CT>
int run_encode_t(const float *x, long B, const void *prepared, float lscale, int N, int K, int D, int iters,
                 uint8_t *out_u8, int64_t *out_i64, void *workspace, size_t workspace_bytes, hipStream_t st,
                 Prof *prof, const int64_t *init_idx, unsigned flags, float *logits_out,
                 uint8_t *codes_also /* with out_i64: the same indexes as unpacked bytes [B][N] */) {
    g_last_launches = 0;
    if (!domain_ok(N, K, D)) return domain_err(N, K, D);
    if (B < 0 || iters < 0 || iters > 60 || (out_u8 == nullptr) == (out_i64 == nullptr)) return MCQ_EINVAL;
    // entries of more than 256-entry codebooks do not fit the byte outputs (encode(as_bytes=True) asserts the same, :271)
    if (sizeof(CT) > 1 && (out_u8 != nullptr || codes_also != nullptr)) return MCQ_EINVAL;
    if (B == 0) return 0;
    if (!x || !prepared || !workspace) return MCQ_EINVAL;
    const size_t per = workspace_per_vector(N, K, D), slack = workspace_slack(D);
    if (workspace_bytes < slack + per) return MCQ_EWORKSPACE;
    long chunk = (long)((workspace_bytes - slack) / per);
    if (chunk > B) chunk = B;
    if (chunk < B && chunk < 128) return MCQ_EWORKSPACE;
    if (chunk < B) chunk &= ~127L;
    const Prepared P = prepared_view(prepared, N, K, D);
    const int pack = (out_u8 != nullptr && K == 16 && N >= 2) ? 2 : 1;
    // fixed-point skipping (opt-in): vectors whose indexes a pass leaves unchanged drop out of the later
    // passes (k_compact); results are identical, the cost becomes data dependent
    const bool skip = (flags & MCQ_ENCODE_SKIP_FIXED_POINTS) != 0 && iters >= 2;

    for (long lo = 0; lo < B; lo += chunk) {
        const long Bc = (B - lo < chunk) ? (B - lo) : chunk;
        const WorkspaceT<CT> w = carve<CT>(workspace, Bc, N, K, D);
        const int xh = (flags & MCQ_ENCODE_X_FP16) ? 1 : 0;   // rows of 2-byte elements
        const float *xc = xh ? reinterpret_cast<const float *>(reinterpret_cast<const uint16_t *>(x) + lo * D) : x + lo * D;
        int rc;
        // the frames as limb planes (and |x|^2), once per call: both products of the call read them
        // the frames as limb planes, centered (x - mean: both products of the call read them; |x - mean|^2 rides along)
        if (init_idx == nullptr || iters > 0) {
            if (prof) prof->begin(CAT_XX);
            rc = launch_fix_rows(xc, xh, Bc, D, D, w.xf, w.xe, w.xx, st, P.mean);
            if (rc) return rc;
            if (prof) prof->end(CAT_XX);
        }
        if (init_idx != nullptr) {
            hipLaunchKernelGGL(k_import_indexes<CT>, dim3((unsigned)((Bc * N + 255) / 256)), dim3(256), 0, st,
                               init_idx + lo * N, Bc * N, K, w.idx);
            MCQ_LAUNCH_CHECK();
            if (prof) prof->untimed();
        } else {
            if (prof) prof->begin(CAT_LOGITS);
            rc = launch_logits(w.xf, w.xe, Bc, P, N, K, D, lscale,
                               (flags & MCQ_ENCODE_LSCALE_FROM_PREPARED) ? P.scales + 1 : nullptr,
                               logits_out ? logits_out + lo * N * K : nullptr, w.idx, st);
            if (rc) return rc;
            if (prof) prof->end(CAT_LOGITS);
        }
        if (iters > 0) {   // what the passes read per vector: the x.C products, once per call
            if (prof) prof->begin(CAT_XC);
            rc = launch_xc(w.xf, w.xe, Bc, P.Cf, P.Ce, (long)N * K, D, w.XC, st);
            if (rc) return rc;
            if (prof) prof->end(CAT_XC);
        }
        int iters_left = iters;
        // without skipping: indexes are refined in place in w.idx, nothing is packed
        CT *idx_cur = w.idx, *idx_new = skip ? w.idxB : w.idx, *idx_pk = w.idxC;
        const int *map_cur = nullptr, *nact = nullptr;
        int *map_nxt = w.map[0], *map_spare = w.map[1];
        if (skip) {
            hipError_t e = hipMemsetAsync(w.cnt, 0, 64 * sizeof(int), st);
            if (e != hipSuccess) return (int)e;
            if (prof) prof->untimed();
        }
        bool wrote_direct = false;
        if constexpr (sizeof(CT) == 1) {
            // 16 or 8 codebooks of 16 entries (the trainer's first phase at 8 / 4 bytes per frame): ALL passes of the call in one launch of
            // persistent workgroups that hold the Gram matrix in LDS (mcq_pass16_kernels.h); not under the profiler, whose
            // categories are the separate launches
            if (K == 16 && (N == 16 || N == 8) && iters > 0 && !skip && prof == nullptr && pass16_enabled()) {
                // (the dynamic LDS above 64 KB has to be allowed once per DEVICE: a process may drive several)
                static bool allowed16[64] = {};
                int dev = 0;
                if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 63;
                if (!allowed16[dev] || dev == 63) {
                    hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tf_pass16<16>),
                                                          hipFuncAttributeMaxDynamicSharedMemorySize, p16_lds_bytes<16>());
                    if (attr == hipSuccess)
                        attr = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tf_pass16<8>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, p16_lds_bytes<8>());
                    if (attr != hipSuccess) return (int)attr;
                    allowed16[dev] = true;
                }
                Pass16Args a;
                a.G = P.G; a.XC = w.XC; a.xx = w.xx; a.Q = P.Q; a.idx = w.idx; a.B = Bc; a.iters = iters;
                const bool direct_out = pack == 1;
                a.out_i64 = (direct_out && out_i64) ? out_i64 + lo * N : nullptr;
                a.out_u8 = direct_out ? (out_u8 ? out_u8 + lo * N : (codes_also ? codes_also + lo * N : nullptr)) : nullptr;
                const long wgs = (Bc + kP16Waves - 1) / kP16Waves;
                if (N == 16)   // one workgroup per CU (157,696 B of LDS), two with eight codebooks (62,464 B)
                    hipLaunchKernelGGL(k_tf_pass16<16>, dim3((unsigned)(wgs < 256 ? wgs : 256)), dim3(64 * kP16Waves), p16_lds_bytes<16>(), st, a);
                else
                    hipLaunchKernelGGL(k_tf_pass16<8>, dim3((unsigned)(wgs < 512 ? wgs : 512)), dim3(64 * kP16Waves), p16_lds_bytes<8>(), st, a);
                MCQ_LAUNCH_CHECK();
                if (direct_out) continue;
                iters_left = 0;
            }
        }
        // E / R of pass it + 1 can be formed by the wave that emits the indexes of pass it (tf_emit), which saves that pass
        // its two E / R launches: 4, 8 or 16 codebooks, no compaction of the vectors between the passes, not under the profiler
        const bool er_in_emit = !skip && (N == 4 || N == 8 || N == 16);
        bool er_ready = false;
        for (int it = 0; it < iters_left; ++it) {
            if (!er_ready) {
                if (prof) prof->begin(CAT_ER);
                rc = launch_tf_er<CT>(N, P.G, w.XC, idx_cur, w.xx, Bc, K, w.E, w.R, w.gterms, nact, map_cur, st);
                if (rc) return rc;
                if (prof) prof->end(CAT_ER);
            }
            er_ready = false;
            if (prof) prof->begin(CAT_STAGE0);
            rc = launch_tf_stage0(K, N, P.G, w.XC, idx_cur, w.R, P.Q, Bc, (N == 1) ? 1 : w.tf.kc[0], reinterpret_cast<CT *>(w.tf.ent),
                                  w.tf.S[0], (N == 1) ? idx_new : static_cast<CT *>(nullptr), nact, map_cur, st);
            if (rc) return rc;
            if (prof) prof->end(CAT_STAGE0);
            if (N >= 2) {
                // last pass, nothing to pack or to scatter back: the winners go straight to the caller's arrays (tf_emit)
                TfLists L = w.tf;
                if (er_in_emit && it + 1 < iters) {
                    L.erG = P.G; L.erXC = w.XC; L.erxx = w.xx; L.erE = w.E; L.erR = w.R; L.erK = K;
                    er_ready = true;
                }
                const bool direct_out = (it + 1 == iters) && !skip && pack == 1;
                if (direct_out) {
                    L.out_i64 = out_i64 ? out_i64 + lo * N : nullptr;
                    L.out_u8 = out_u8 ? out_u8 + lo * N : (codes_also ? codes_also + lo * N : nullptr);
                    wrote_direct = true;
                }
                rc = run_tf_combines<CT>(P.G, idx_cur, idx_new, w, L, Bc, N, K, nact, st, prof);
                if (rc) return rc;
            }
            if (skip) {
                const int last = (it + 1 == iters) ? 1 : 0;
                if (prof) prof->begin(CAT_TAIL);
                hipLaunchKernelGGL(k_compact<CT>, dim3((unsigned)((Bc + 255) / 256)), dim3(256), 0, st, idx_cur, idx_new,
                                   map_cur, nact, Bc, N, last, w.final_idx, idx_pk, map_nxt, w.cnt + it);
                MCQ_LAUNCH_CHECK();
                if (prof) prof->end(CAT_TAIL);
                // rotate: the packed list becomes the current one
                CT *t = idx_cur; idx_cur = idx_pk; idx_pk = t;
                int *old_map = const_cast<int *>(map_cur);
                map_cur = map_nxt;
                map_nxt = old_map ? old_map : map_spare;
                nact = w.cnt + it;
            }
        }
        if (wrote_direct) continue;
        const CT *result = (skip && iters > 0) ? w.final_idx : w.idx;
        const long outn = (out_i64 != nullptr) ? Bc * N : Bc * (N / pack);
        if (prof) prof->begin(CAT_TAIL);
        hipLaunchKernelGGL(k_finalize<CT>, dim3((unsigned)((outn + 255) / 256)), dim3(256), 0, st, result, Bc, N, pack,
                           out_u8 ? out_u8 + lo * (N / pack) : nullptr, out_i64 ? out_i64 + lo * N : nullptr,
                           codes_also ? codes_also + lo * N : nullptr);
        MCQ_LAUNCH_CHECK();
        if (prof) prof->end(CAT_TAIL);
    }
    return 0;
}

int run_encode(const float *x, long B, const void *prepared, float lscale, int N, int K, int D, int iters,
               uint8_t *out_u8, int64_t *out_i64, void *workspace, size_t workspace_bytes, hipStream_t st,
               Prof *prof, const int64_t *init_idx = nullptr, unsigned flags = 0, float *logits_out = nullptr,
               uint8_t *codes_also = nullptr) {
    if (K > 256)
        return run_encode_t<uint16_t>(x, B, prepared, lscale, N, K, D, iters, out_u8, out_i64, workspace, workspace_bytes, st, prof,
                                      init_idx, flags, logits_out, codes_also);
    return run_encode_t<uint8_t>(x, B, prepared, lscale, N, K, D, iters, out_u8, out_i64, workspace, workspace_bytes, st, prof,
                                 init_idx, flags, logits_out, codes_also);
}

// floats per lane of k_decode_backward when every row involved is 16-byte aligned.  Codebooks of 64 entries and more: 4 (a
// row has few matching vectors, the kernel is bound by scanning the index column, so as few waves per row as possible:
// 22.5 us with 4, 28.3 with 2, 41.4 with 1 at 8 x 256, dim 512, 4,096 vectors).  Smaller codebooks: the widest of 4, 2, 1
// that leaves four feature chunks (a row gathers many vectors; more, narrower waves and an L2 that holds a quarter of the
// gradient matrix: 35.5 us with 4, 28.7 with 2, 30.2 with 1 at 16 x 16).  1 for unaligned rows.
int db_cw_of(int D, int K) {
    if ((D & 3) != 0) return 1;
    if (K >= 64) return 4;
    return D >= 1024 ? 4 : (D >= 512 ? 2 : 1);
}
int db_cw(const void *g, const void *out, int D, int K, long gsb, long gsn, const void *dotw = nullptr) {
    const bool al = ((D & 3) == 0) && ((gsb & 3) == 0) && ((gsn & 3) == 0) && ((reinterpret_cast<uintptr_t>(g) & 15) == 0) &&
                    ((reinterpret_cast<uintptr_t>(out) & 15) == 0) && ((reinterpret_cast<uintptr_t>(dotw) & 15) == 0);
    return al ? db_cw_of(D, K) : 1;
}
int db_chunks(int D, int cw) { return (D + 64 * cw - 1) / (64 * cw); }

template <typename IdxT>
int launch_decode_backward(const float *g, const IdxT *idx, long B, int N, int K, int D, float *out, long gsb, long gsn,
                           int idx_stride, hipStream_t st, const float *sa = nullptr, const float *sb = nullptr, float sc = 1.0f,
                           const float *dotw = nullptr, float *dot_part = nullptr) {
    const int cw = db_cw(g, out, D, K, gsb, gsn, dotw), chunks = db_chunks(D, cw);
    const long rowgroups = ((long)N * K + 3) / 4;
    long blocks;
    if (chunks <= 8 && (8 % chunks) == 0) blocks = ((rowgroups + (8 / chunks) - 1) / (8 / chunks)) * 8;
    else blocks = ((long)N * K * chunks + 3) / 4;
    const dim3 grid((unsigned)blocks), block(256);
    if (cw == 4) hipLaunchKernelGGL((k_decode_backward<IdxT, 4>), grid, block, 0, st, g, idx, B, N, K, D, chunks, out, gsb, gsn, idx_stride, sa, sb, sc, dotw, dot_part);
    else if (cw == 2) hipLaunchKernelGGL((k_decode_backward<IdxT, 2>), grid, block, 0, st, g, idx, B, N, K, D, chunks, out, gsb, gsn, idx_stride, sa, sb, sc, dotw, dot_part);
    else hipLaunchKernelGGL((k_decode_backward<IdxT, 1>), grid, block, 0, st, g, idx, B, N, K, D, chunks, out, gsb, gsn, idx_stride, sa, sb, sc, dotw, dot_part);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}
}  // namespace

extern "C" {

int mcq_abi_version(void) { return MCQ_ABI_VERSION; }

int mcq_padded_dim(int D) { return round_up16(D); }

size_t mcq_prepared_bytes(int N, int K, int D) {
    if (N <= 0 || K <= 0 || D <= 0) return 0;
    return prepared_layout(N, K, D).total;
}

static int prepare_impl(const float *centers, float cscale_exp, const float *scales_dev, const float *weight,
                        const float *bias, int N, int K, int D, void *prepared, void *stream, const float *raw_cs = nullptr,
                        const float *raw_ls = nullptr, float speed = 0.f, float *scales_out2 = nullptr) {
    if (!domain_ok(N, K, D)) return domain_err(N, K, D);
    if (!centers || !prepared || ((weight == nullptr) != (bias == nullptr))) return MCQ_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const PreparedLayout l = prepared_layout(N, K, D);
    char *b = static_cast<char *>(prepared);
    const long rows = (long)N * K;
    const int Dp = round_up16(D);
    const unsigned grid = (unsigned)((rows + 3) / 4);
    hipLaunchKernelGGL(k_prepare_rows, dim3(grid), dim3(256), 0, st, centers, cscale_exp, 1, rows, D, Dp,
                       reinterpret_cast<float *>(b + l.offC), static_cast<float *>(nullptr) /* Q: of the centered rows, below */, scales_dev,
                       (scales_dev || raw_cs) ? reinterpret_cast<float *>(b + l.offScales) : static_cast<float *>(nullptr), raw_cs,
                       raw_ls, speed, scales_out2);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    if (!weight) return 0;      // decode only: the scaled centers are all mcq_decode reads (mcq_prepared_decode_bytes)
    const float *C = reinterpret_cast<const float *>(b + l.offC);
    float *cmean = reinterpret_cast<float *>(b + l.offCMean);
    hipLaunchKernelGGL(k_centers_mean, dim3((unsigned)(Dp / 16)), dim3(1024), 0, st, C, N, K, Dp,
                       reinterpret_cast<float *>(b + l.offMean), cmean);
    e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    // the scaled centers and the classifier rows as limb planes (the tables of the fixed-point products), one launch; the
    // bias rides along
    // (the centers enter every table of the search with their codebook's mean taken out -- oracle "TABLE FORM", centering; their
    // sums of squares Q come out of the same pass)
    int rc = launch_fix_rows2(fix_rows_args(C, 0, rows, Dp, Dp, reinterpret_cast<int8_t *>(b + l.offCf), reinterpret_cast<int *>(b + l.offCe),
                                            reinterpret_cast<float *>(b + l.offQ), nullptr, nullptr, cmean, K, Dp),
                              fix_rows_args(weight, 0, rows, D, D, reinterpret_cast<int8_t *>(b + l.offWf), reinterpret_cast<int *>(b + l.offWe),
                                            nullptr, bias, reinterpret_cast<float *>(b + l.offBias), nullptr, 0, 0,
                                            reinterpret_cast<const float *>(b + l.offMean), reinterpret_cast<float *>(b + l.offWmu)), st);
    if (rc) return rc;
    {
        // Gram matrix of the scaled centers: the x.C product with the centers themselves as the frames
        rc = launch_xc(reinterpret_cast<const int8_t *>(b + l.offCf), reinterpret_cast<const int *>(b + l.offCe), rows,
                       reinterpret_cast<const int8_t *>(b + l.offCf), reinterpret_cast<const int *>(b + l.offCe), rows, D,
                       reinterpret_cast<float *>(b + l.offG), st);
        if (rc) return rc;
    }
    return 0;
}

size_t mcq_prepared_decode_bytes(int N, int K, int D) {
    if (N <= 0 || K <= 0 || D <= 0) return 0;
    return prepared_layout(N, K, D).offCf;      // scaled centers + their sums of squares
}

size_t mcq_prepared_mean_offset(int N, int K, int D) {
    if (N <= 0 || K <= 0 || D <= 0) return 0;
    return prepared_layout(N, K, D).offMean;
}

int mcq_prepare(const float *centers, float cscale_exp, const float *weight, const float *bias, int N, int K, int D,
                void *prepared, void *stream) {
    return prepare_impl(centers, cscale_exp, nullptr, weight, bias, N, K, D, prepared, stream);
}

int mcq_prepare_dev(const float *centers, const float *scales_exp, const float *weight, const float *bias, int N,
                    int K, int D, void *prepared, void *stream) {
    if (!scales_exp) return MCQ_EINVAL;
    return prepare_impl(centers, 1.0f, scales_exp, weight, bias, N, K, D, prepared, stream);
}

int mcq_prepare_params(const float *centers, const float *centers_scale, const float *logits_scale, float speed,
                       const float *weight, const float *bias, int N, int K, int D, void *prepared, float *scales_exp_out,
                       void *stream) {
    if (!centers_scale || !logits_scale) return MCQ_EINVAL;
    return prepare_impl(centers, 1.0f, nullptr, weight, bias, N, K, D, prepared, stream, centers_scale, logits_scale, speed,
                        scales_exp_out);
}

size_t mcq_encode_workspace_bytes(long B, int N, int K, int D) {
    if (B <= 0 || N <= 0 || K <= 0 || D <= 0 || !domain_ok(N, K, D)) return 48 * 256;
    const long dc = default_chunk(N, K, D), chunk = B < dc ? B : dc;
    return workspace_slack(D) + workspace_per_vector(N, K, D) * (size_t)chunk;
}

int mcq_encode(const float *x, long B, const void *prepared, float lscale_exp, int N, int K, int D, int refine_iters,
               uint8_t *out_u8, int64_t *out_i64, void *workspace, size_t workspace_bytes, void *stream) {
    return run_encode(x, B, prepared, lscale_exp, N, K, D, refine_iters, out_u8, out_i64, workspace,
                      workspace_bytes, static_cast<hipStream_t>(stream), nullptr);
}

int mcq_encode_ex(const float *x, long B, const void *prepared, float lscale_exp, int N, int K, int D,
                  int refine_iters, uint8_t *out_u8, int64_t *out_i64, void *workspace, size_t workspace_bytes,
                  void *stream, unsigned flags) {
    return run_encode(x, B, prepared, lscale_exp, N, K, D, refine_iters, out_u8, out_i64, workspace,
                      workspace_bytes, static_cast<hipStream_t>(stream), nullptr, nullptr, flags);
}

int mcq_refine_indexes(const float *x, long B, const void *prepared, int N, int K, int D, int refine_iters,
                       const int64_t *idx_in, int64_t *idx_out, void *workspace, size_t workspace_bytes,
                       void *stream) {
    if (B > 0 && (!idx_in || !idx_out)) return MCQ_EINVAL;
    return run_encode(x, B, prepared, 1.0f, N, K, D, refine_iters, nullptr, idx_out, workspace, workspace_bytes,
                      static_cast<hipStream_t>(stream), nullptr, B > 0 ? idx_in : nullptr);
}

int mcq_decode(const void *codes, int code_bytes, int codes_per_row, long B, const void *prepared, int N, int K, int D,
               float *out, void *stream) {
    if (!domain_ok(N, K, D)) return domain_err(N, K);
    if (B < 0 || codes_per_row <= 0 || N % codes_per_row != 0) return MCQ_EINVAL;
    const int rep = N / codes_per_row;
    if (!(rep == 1 || rep == 2 || rep == 4 || rep == 8 || rep == 16)) return MCQ_EINVAL;
    if (code_bytes != 1 && code_bytes != 8) return MCQ_EINVAL;
    if (B == 0) return 0;
    if (!codes || !prepared || !out) return MCQ_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const Prepared P = prepared_view(prepared, N, K, D);
    const int Dp = round_up16(D);
    const unsigned grid = (unsigned)((B + 3) / 4);
    const int J = (Dp / 4 + 63) / 64;
    // XCD-sliced kernel (unpacked codes, batches big enough to fill the chip; MCQ_DECODE_SLICED=0 disables: tuning hook)
    static const bool sliced_ok = !(getenv("MCQ_DECODE_SLICED") && atoi(getenv("MCQ_DECODE_SLICED")) == 0);
    // LDS-resident kernel: batches of >= 16,384 vectors whose codebook slice (N*K*64 B) fits the LDS
    // (N >= 8: with fewer rows per vector the L2 gathers of the sliced kernel measured faster; 36.9 vs 52.2 us at 8 x 256, 65,536 vectors)
    const char *lds_env = getenv("MCQ_DECODE_LDS_MIN");   // test / tuning hook, read per call
    const long lds_min_b = lds_env ? atol(lds_env) : 16384;
    // block-staged LDS-resident kernel (k_decode_blk): packed byte codes, 4, 8 or 16 per vector, 64-byte slices when they fit the
    // LDS beside the two code buffers, 32-byte slices otherwise (16 x 256).  MCQ_DECODE_BLK=0: the other kernels (tuning hook)
    {
        const char *blk_env = getenv("MCQ_DECODE_BLK");
        const int blk_mode = blk_env ? atoi(blk_env) : 1;
        const size_t lds_max = 160 * 1024, cbytes = 2 * 16384;
        int lpv = 0;
        if ((size_t)N * K * 64 + cbytes <= lds_max) lpv = 4;
        else if ((size_t)N * K * 32 + cbytes <= lds_max) lpv = 2;
        if (blk_mode != 0 && sliced_ok && rep == 1 && code_bytes == 1 && B >= lds_min_b && K >= 32 && (D & 3) == 0 && lpv != 0 &&
            (N == 4 || N == 8 || N == 16) && ((reinterpret_cast<uintptr_t>(codes) & 15) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0)) {
            const int W = 4 * lpv, ns = Dp / W, per_xcd = (ns + 7) / 8;
            int groups = 256 / (8 * per_xcd);            // one workgroup per CU
            groups = groups < 1 ? 1 : groups;
            const unsigned g = (unsigned)(8 * per_xcd * groups);
            const long per = (((B + groups - 1) / groups) + 255) / 256 * 256;
            const size_t lds = (size_t)N * K * W * 4 + cbytes;
            const uint8_t *cp = static_cast<const uint8_t *>(codes);
#define MCQ_DECB(NN, LL)                                                                                                    \
    do {                                                                                                                    \
        static bool allowed[64] = {};                                                                                       \
        int dev = 0;                                                                                                        \
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 63;                                             \
        if (!allowed[dev] || dev == 63) {                                                                                   \
            const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_decode_blk<NN, LL>),              \
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);          \
            if (attr != hipSuccess) return (int)attr;                                                                       \
            allowed[dev] = true;                                                                                            \
        }                                                                                                                   \
        hipLaunchKernelGGL((k_decode_blk<NN, LL>), dim3(g), dim3(1024), lds, st, cp, B, P.C, K, D, Dp, groups, per, out);   \
    } while (0)
            if (N == 4) { if (lpv == 4) MCQ_DECB(4, 4); else MCQ_DECB(4, 2); }
            else if (N == 8) { if (lpv == 4) MCQ_DECB(8, 4); else MCQ_DECB(8, 2); }
            else { if (lpv == 4) MCQ_DECB(16, 4); else MCQ_DECB(16, 2); }
#undef MCQ_DECB
            hipError_t e6 = hipGetLastError();
            return e6 == hipSuccess ? 0 : (int)e6;
        }
    }
    if (sliced_ok && rep == 1 && B >= lds_min_b && (size_t)N * K * 64 <= 144 * 1024 && K >= 32 && N >= 8) {
        const int ns = Dp / 16, per_xcd = (ns + 7) / 8;
        int groups = 256 / (8 * per_xcd);            // one workgroup per CU
        groups = groups < 1 ? 1 : groups;
        const unsigned g = (unsigned)(8 * per_xcd * groups);
        const size_t lds = (size_t)N * K * 64;
        if (code_bytes == 1)
            hipLaunchKernelGGL((k_decode_lds<uint8_t>), dim3(g), dim3(1024), lds, st, static_cast<const uint8_t *>(codes), B,
                               P.C, N, K, D, Dp, groups, out);
        else
            hipLaunchKernelGGL((k_decode_lds<int64_t>), dim3(g), dim3(1024), lds, st, static_cast<const int64_t *>(codes), B,
                               P.C, N, K, D, Dp, groups, out);
        hipError_t e4 = hipGetLastError();
        return e4 == hipSuccess ? 0 : (int)e4;
    }
    if (sliced_ok && rep == 1 && B >= 4096 && K >= 32) {   // (16-entry codebooks: the per-vector kernels measured faster)
        int lpv = 4;
        while (lpv * 32 < Dp) lpv *= 2;          // 8 slices x lpv lanes x 4 floats cover Dp
        if (lpv <= 64) {
            const int vpw = 64 / lpv;
            const unsigned g = (unsigned)(((B + 4 * vpw - 1) / (4 * vpw)) * 8);
#define MCQ_DECS_LAUNCH(T, CHH, LL)                                                                              \
    hipLaunchKernelGGL((k_decode_sliced<T, CHH, LL>), dim3(g), dim3(256), 0, st, static_cast<const T *>(codes), B, P.C, N, \
                       K, D, Dp, out)
#define MCQ_DECS_LPV(T, CHH)                                                                                     \
    switch (lpv) {                                                                                               \
        case 4: MCQ_DECS_LAUNCH(T, CHH, 4); break;                                                               \
        case 8: MCQ_DECS_LAUNCH(T, CHH, 8); break;                                                               \
        case 16: MCQ_DECS_LAUNCH(T, CHH, 16); break;                                                             \
        case 32: MCQ_DECS_LAUNCH(T, CHH, 32); break;                                                             \
        default: MCQ_DECS_LAUNCH(T, CHH, 64); break;                                                             \
    }
            if (code_bytes == 1) {
                if (N <= 4) { MCQ_DECS_LPV(uint8_t, 4) } else if (N <= 8) { MCQ_DECS_LPV(uint8_t, 8) } else { MCQ_DECS_LPV(uint8_t, 16) }
            } else {
                if (N <= 4) { MCQ_DECS_LPV(int64_t, 4) } else if (N <= 8) { MCQ_DECS_LPV(int64_t, 8) } else { MCQ_DECS_LPV(int64_t, 16) }
            }
#undef MCQ_DECS_LPV
#undef MCQ_DECS_LAUNCH
            hipError_t e3 = hipGetLastError();
            return e3 == hipSuccess ? 0 : (int)e3;
        }
    }
#define MCQ_DEC_CASE(NN, JJ)                                                                                    \
    if (code_bytes == 1 && rep == 1 && N == NN && J == JJ) {                                                    \
        hipLaunchKernelGGL((k_decode_reg<NN, JJ>), dim3(grid), dim3(256), 0, st, static_cast<const uint8_t *>(codes), \
                           B, P.C, K, D, Dp, out);                                                              \
        hipError_t e2 = hipGetLastError();                                                                      \
        return e2 == hipSuccess ? 0 : (int)e2;                                                                  \
    }
    MCQ_DEC_CASE(8, 2)
    MCQ_DEC_CASE(8, 1)
    MCQ_DEC_CASE(4, 1)
    MCQ_DEC_CASE(4, 2)
    MCQ_DEC_CASE(4, 4)
    MCQ_DEC_CASE(16, 1)
    MCQ_DEC_CASE(16, 2)
    MCQ_DEC_CASE(2, 1)
    MCQ_DEC_CASE(2, 2)
#undef MCQ_DEC_CASE
    if (code_bytes == 1)
        hipLaunchKernelGGL((k_decode<uint8_t>), dim3(grid), dim3(256), 0, st, static_cast<const uint8_t *>(codes),
                           codes_per_row, B, P.C, N, K, D, Dp, out);
    else
        hipLaunchKernelGGL((k_decode<int64_t>), dim3(grid), dim3(256), 0, st, static_cast<const int64_t *>(codes),
                           codes_per_row, B, P.C, N, K, D, Dp, out);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

int mcq_decode_backward(const float *grad_out, const int64_t *idx, long B, int N, int K, int D, float *gC,
                        void *stream) {
    if (N <= 0 || K <= 0 || D <= 0 || B < 0 || !gC) return MCQ_EINVAL;
    if (B > 0 && (!grad_out || !idx)) return MCQ_EINVAL;
    return launch_decode_backward<int64_t>(grad_out, idx, B, N, K, D, gC, (long)D, 0L, N, static_cast<hipStream_t>(stream));
}

int mcq_decode_backward_u8(const float *grad_out, const uint8_t *codes, long B, int N, int K, int D, float *gC,
                           void *stream) {
    if (K > 256) return MCQ_EUNSUPPORTED;       // (byte codes: the trainer's entry points stay at K <= 256, include/mcq.h)
    if (N <= 0 || K <= 0 || D <= 0 || B < 0 || !gC) return MCQ_EINVAL;
    if (B > 0 && (!grad_out || !codes)) return MCQ_EINVAL;
    return launch_decode_backward<uint8_t>(grad_out, codes, B, N, K, D, gC, (long)D, 0L, N, static_cast<hipStream_t>(stream));
}

int mcq_scatter_rows(const float *grad, long stride_b, long stride_n, const int64_t *idx, int idx_stride, long B, int N,
                     int K, int D, float *out, void *stream) {
    if (N <= 0 || K <= 0 || D <= 0 || B < 0 || !out || idx_stride < N) return MCQ_EINVAL;
    if (B > 0 && (!grad || !idx)) return MCQ_EINVAL;
    return launch_decode_backward<int64_t>(grad, idx, B, N, K, D, out, stride_b, stride_n, idx_stride, static_cast<hipStream_t>(stream));
}

int mcq_jcl_prefix_fwd(const float *hp, const float *emb, const int64_t *idx, long B, int N, int K, int H, float scale,
                       float *A, void *stream) {
    if (N < 2 || K < 1 || H < 1 || B < 0) return MCQ_EINVAL;
    if (B == 0) return 0;
    if (!hp || !emb || !idx || !A) return MCQ_EINVAL;
    hipLaunchKernelGGL(k_jcl_prefix_fwd, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream), hp,
                       emb, idx, B, N, K, H, scale, A);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

int mcq_jcl_prefix_bwd(const float *A, const float *gA, long B, int N, int H, float scale, float *g_hp, float *gE,
                       void *stream) {
    if (N < 2 || H < 1 || B < 0) return MCQ_EINVAL;
    if (B == 0) return 0;
    if (!A || !gA || !g_hp || !gE) return MCQ_EINVAL;
    hipLaunchKernelGGL(k_jcl_prefix_bwd, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream), A,
                       gA, B, N, H, scale, g_hp, gE);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

namespace {
// workspace of the logits entry points: arg max bytes, then the frames as limb planes and their exponents
struct LogitsWs {
    uint8_t *idx8;
    int8_t *xf;
    int *xe;
    size_t total;
};
LogitsWs logits_ws(void *ws, long B, int N, int D) {
    char *p = static_cast<char *>(ws);
    LogitsWs w;
    size_t off = 0;
    w.idx8 = reinterpret_cast<uint8_t *>(p + off);
    off = align256(off + (size_t)B * N);
    w.xf = reinterpret_cast<int8_t *>(p + off);
    off = align256(off + fix_plane_bytes(B, D));
    w.xe = reinterpret_cast<int *>(p + off);
    off = align256(off + (size_t)fix_round_rows(B) * 4);
    w.total = off;
    return w;
}
}  // namespace

size_t mcq_logits_workspace_bytes(long B, int N, int D) {
    if (B <= 0 || N <= 0 || D <= 0) return 256;
    return logits_ws(nullptr, B, N, D).total;
}

int mcq_logits(const float *x, long B, const void *prepared, float lscale_exp, int N, int K, int D, float *out,
               void *workspace, size_t workspace_bytes, void *stream) {
    if (!domain_ok(N, K, D)) return MCQ_EUNSUPPORTED;
    if (B == 0) return 0;
    if (!x || !prepared || !out || B < 0 || !workspace) return MCQ_EINVAL;
    if (workspace_bytes < mcq_logits_workspace_bytes(B, N, D)) return MCQ_EWORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const Prepared P = prepared_view(prepared, N, K, D);
    const LogitsWs w = logits_ws(workspace, B, N, D);
    int rc = launch_fix_rows(x, 0, B, D, D, w.xf, w.xe, nullptr, st, P.mean);      // (the logits product reads centered frames: FixGemm::wmu)
    if (rc) return rc;
    return launch_logits(w.xf, w.xe, B, P, N, K, D, lscale_exp, nullptr, out, nullptr, st);
}

// ------------------------------------------------------------------ trainer pieces
int mcq_logits_argmax(const float *x, long B, const void *prepared, float lscale_exp, int N, int K, int D,
                      float *logits_out, int64_t *argmax_out, void *workspace, size_t workspace_bytes, void *stream,
                      unsigned flags) {
    if (!domain_ok(N, K, D) || K > 256) return MCQ_EUNSUPPORTED;      // (the trainer's entry point: one-byte entries)
    if (B < 0) return MCQ_EINVAL;
    if (B == 0) return 0;
    if (!x || !prepared || !logits_out || !argmax_out || !workspace) return MCQ_EINVAL;
    if (workspace_bytes < mcq_logits_workspace_bytes(B, N, D)) return MCQ_EWORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const Prepared P = prepared_view(prepared, N, K, D);
    const LogitsWs w = logits_ws(workspace, B, N, D);
    int rc = launch_fix_rows(x, (flags & MCQ_ENCODE_X_FP16) ? 1 : 0, B, D, D, w.xf, w.xe, nullptr, st, P.mean);
    if (rc) return rc;
    rc = launch_logits(w.xf, w.xe, B, P, N, K, D, lscale_exp,
                       (flags & MCQ_ENCODE_LSCALE_FROM_PREPARED) ? P.scales + 1 : nullptr, logits_out, w.idx8, st);
    if (rc) return rc;
    hipLaunchKernelGGL(k_export_indexes, dim3((unsigned)((B * N + 255) / 256)), dim3(256), 0, st, w.idx8, B * N, argmax_out);
    MCQ_LAUNCH_CHECK();
    return 0;
}

// logits (stored) + arg max + refinement passes in one call: the frames become limb planes once, the indexes stay bytes
// until the end (what QuantizerTrainer.step runs: mcq_logits_argmax followed by mcq_refine_indexes, without the detours)
int mcq_logits_refine(const float *x, long B, const void *prepared, float lscale_exp, int N, int K, int D, int refine_iters,
                      float *logits_out, int64_t *idx_out, void *workspace, size_t workspace_bytes, void *stream,
                      unsigned flags) {
    return mcq_logits_refine_codes(x, B, prepared, lscale_exp, N, K, D, refine_iters, logits_out, idx_out, nullptr, workspace,
                                   workspace_bytes, stream, flags);
}

int mcq_logits_refine_codes(const float *x, long B, const void *prepared, float lscale_exp, int N, int K, int D, int refine_iters,
                            float *logits_out, int64_t *idx_out, uint8_t *codes_out, void *workspace, size_t workspace_bytes,
                            void *stream, unsigned flags) {
    if (K > 256) return MCQ_EUNSUPPORTED;       // a trainer entry point (stored logits for the fused loss kernels): K <= 256
    if (B > 0 && (!logits_out || !idx_out)) return MCQ_EINVAL;
    return run_encode(x, B, prepared, lscale_exp, N, K, D, refine_iters, nullptr, idx_out, workspace, workspace_bytes,
                      static_cast<hipStream_t>(stream), nullptr, nullptr, flags & ~MCQ_ENCODE_SKIP_FIXED_POINTS, logits_out,
                      codes_out);
}

namespace {
long loss_rows_per_chunk(long B) {   // at most ~1024 chunks, at least 64 rows each
    long r = (B + 1023) / 1024;
    r = r < 64 ? 64 : r;
    return (r + 15) / 16 * 16;
}
long loss_chunks(long B) { const long r = loss_rows_per_chunk(B); return (B + r - 1) / r; }
}  // namespace

size_t mcq_loss_workspace_bytes(long B, int N, int K) {
    if (B <= 0) return 256;
    return (size_t)loss_chunks(B) * N * (2 * (size_t)K + 1) * sizeof(float) + 256;
}

int mcq_loss_fwd(const float *logits, const int64_t *idx, long B, int N, int K, float *lse, float *chosen_sum,
                 float *prob_sum, float *count, void *workspace, size_t workspace_bytes, void *stream) {
    if (!is_pow2(K) || K < 16 || K > 256 || N < 1) return MCQ_EUNSUPPORTED;
    if (B <= 0) return MCQ_EINVAL;
    if (!logits || !idx || !lse || !chosen_sum || !prob_sum || !count || !workspace) return MCQ_EINVAL;
    if (workspace_bytes < mcq_loss_workspace_bytes(B, N, K)) return MCQ_EWORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long rpc = loss_rows_per_chunk(B), chunks = loss_chunks(B);
    float *pp = static_cast<float *>(workspace);
    float *pc = pp + chunks * N * K;
    float *ph = pc + chunks * N * K;
    const dim3 grid((unsigned)(chunks * N)), block(64 * kLossWaves);
#define MCQ_LOSS_CASE(KK)                                                                                             \
    case KK: hipLaunchKernelGGL((k_loss_fwd<KK>), grid, block, 0, st, logits, idx, B, N, rpc, lse, pp, pc, ph); break;
    switch (K) {
        MCQ_LOSS_CASE(16) MCQ_LOSS_CASE(32) MCQ_LOSS_CASE(64) MCQ_LOSS_CASE(128) MCQ_LOSS_CASE(256)
        default: return MCQ_EUNSUPPORTED;
    }
#undef MCQ_LOSS_CASE
    MCQ_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_loss_reduce, dim3((unsigned)N), dim3(256), 0, st, pp, pc, ph, chunks, N, K, prob_sum, count,
                       chosen_sum);
    MCQ_LAUNCH_CHECK();
    return 0;
}

int mcq_loss_bwd(const float *logits, const int64_t *idx, const float *lse, long B, int N, int K, const float *g_chosen,
                 const float *g_prob, float *grad_logits, void *stream) {
    if (!is_pow2(K) || K < 16 || K > 256 || N < 1) return MCQ_EUNSUPPORTED;
    if (B <= 0) return MCQ_EINVAL;
    if (!logits || !idx || !lse || !g_chosen || !g_prob || !grad_logits) return MCQ_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int rpw = K < 64 ? 64 / K : 1;
    const long rows = B * N;
    const dim3 grid((unsigned)((rows + (long)kLossWaves * rpw - 1) / ((long)kLossWaves * rpw))), block(64 * kLossWaves);
#define MCQ_LOSS_CASE(KK)                                                                                             \
    case KK: hipLaunchKernelGGL((k_loss_bwd<KK>), grid, block, 0, st, logits, idx, lse, B, N, g_chosen, g_prob, grad_logits); break;
    switch (K) {
        MCQ_LOSS_CASE(16) MCQ_LOSS_CASE(32) MCQ_LOSS_CASE(64) MCQ_LOSS_CASE(128) MCQ_LOSS_CASE(256)
        default: return MCQ_EUNSUPPORTED;
    }
#undef MCQ_LOSS_CASE
    MCQ_LAUNCH_CHECK();
    return 0;
}

int mcq_loss_tail(const float *sums, const float *prob_sum, const float *count, int N, int K, float entropy_scale,
                  float *losses, float *g, float *g_prob, void *stream) {
    if (!is_pow2(K) || K < 16 || K > 256 || N < 1) return MCQ_EUNSUPPORTED;
    if (!sums || !prob_sum || !count || !losses || !g || !g_prob) return MCQ_EINVAL;
    hipLaunchKernelGGL(k_loss_tail, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), sums, prob_sum, count, N, K,
                       entropy_scale, losses, g, g_prob);
    MCQ_LAUNCH_CHECK();
    return 0;
}

int mcq_recon_fwd(const float *x, const int64_t *idx, long B, const void *prepared, const float *mean, int N, int K,
                  int D, float *err, float *num_part, float *den_part, void *stream) {
    if (!domain_ok(N, K, D) || K > 256) return MCQ_EUNSUPPORTED;
    if (B <= 0) return MCQ_EINVAL;
    if (!x || !idx || !prepared || !mean || !err || !num_part || !den_part) return MCQ_EINVAL;
    const Prepared P = prepared_view(prepared, N, K, D);
    hipLaunchKernelGGL(k_recon_fwd, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream), x, idx, B,
                       P.C, mean, N, K, D, round_up16(D), err, num_part, den_part);
    MCQ_LAUNCH_CHECK();
    return 0;
}


// ------------------------------------------------------------ parameter update
namespace {
// the bf16-piece kernel (k_wgrad_bf3: one workgroup of eight waves per CU, 128 x 128 outputs) takes the large tile-aligned
// products -- the trainer's second phase -- with about one workgroup per CU and at least 512 rows of the batch per split
// (dim 512, 2,048 logits, 4,096 frames: 4 splits 70.8 us per call, 8 splits 78.4, 16 splits 90.9; the fp32 kernel 113.9)
bool wgrad_use_bf3(long B, int M, int D) {
    static const bool f32_only = getenv("MCQ_WGRAD_F32") && atoi(getenv("MCQ_WGRAD_F32")) != 0;      // tuning hook: same sums to fp32 accuracy either way
    return !f32_only && (M % kWbM) == 0 && (D % kWbN) == 0 && M >= 1024 && B >= 2048;
}
int wgrad_splits_bf3(long B, int M, int D) {
    const long tiles = (long)(M / kWbM) * (D / kWbN);
    long s = (256 + tiles / 2) / tiles;
    const long max_s = B / 512;
    s = s > max_s ? max_s : s;
    return (int)(s < 1 ? 1 : s);
}

int wgrad_splits(long B, int M, int D) {
    const long tiles = (long)((M + kWgM - 1) / kWgM) * ((D + kWgN - 1) / kWgN);
    long s = 2048 / tiles;                 // ~8 workgroups per CU
    const long max_s = (B + 127) / 128;    // at least 128 rows of the batch per split
    s = s > max_s ? max_s : s;
    return (int)(s < 1 ? 1 : (s > 64 ? 64 : s));
}
}  // namespace

size_t mcq_weight_grad_workspace_bytes(long B, int M, int D) {
    if (B <= 0 || M <= 0 || D <= 0) return 256;
    return (size_t)wgrad_splits(B, M, D) * ((size_t)M * D + M) * sizeof(float) + 256;
}

int mcq_weight_grad(const float *G, const float *x, long B, int M, int D, const float *scale_dev, float *gW, float *gb,
                    void *workspace, size_t workspace_bytes, void *stream) {
    if (B <= 0 || M <= 0 || D <= 0 || (M & 15) != 0) return MCQ_EINVAL;
    if (!G || !x || !scale_dev || !gW || !gb || !workspace) return MCQ_EINVAL;
    if (workspace_bytes < mcq_weight_grad_workspace_bytes(B, M, D)) return MCQ_EWORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool bf3 = wgrad_use_bf3(B, M, D);
    const int splits = bf3 ? wgrad_splits_bf3(B, M, D) : wgrad_splits(B, M, D);      // (never more than the workspace was sized for)
    long rps = (B + splits - 1) / splits;
    rps = (rps + 31) / 32 * 32;
    float *part = static_cast<float *>(workspace);
    float *partb = part + (size_t)splits * M * D;
    if (bf3) {
        hipLaunchKernelGGL(k_wgrad_bf3, dim3((unsigned)((M / kWbM) * (D / kWbN) * splits)), dim3(512), 0, st, G, x, B, M, D, rps, part, partb);
    } else {
        const unsigned grid = (unsigned)(((M + kWgM - 1) / kWgM) * ((D + kWgN - 1) / kWgN) * splits);
        hipLaunchKernelGGL((k_wgrad_tn<16>), dim3(grid), dim3(256), 0, st, G, x, B, M, D, rps, part, partb);   // (32-row stages: 129 vs 115 us)
    }
    MCQ_LAUNCH_CHECK();
    const long MN = (long)M * D;
    hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)((MN / 4 + 255) / 256)), dim3(256), 0, st, part, partb, splits, MN, M,
                       scale_dev, gW, gb);
    MCQ_LAUNCH_CHECK();
    return 0;
}

int mcq_adam_step(float *p, const float *g, float *m, float *v, long n, double lr, double beta1, double beta2, double eps,
                  double weight_decay, double bias_correction1, double bias_correction2_sqrt, void *stream) {
    if (n < 0 || (n > 0 && (!p || !g || !m || !v))) return MCQ_EINVAL;
    if (n == 0) return 0;
    long blocks = (n / 4 + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
    hipLaunchKernelGGL(k_adam, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), p, g, m, v, n,
                       (float)(lr / bias_correction1), (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps,
                       (float)weight_decay, (float)bias_correction2_sqrt);
    MCQ_LAUNCH_CHECK();
    return 0;
}

int mcq_loss_head_tail(const float *num_part, const float *den_part, long nparts, const float *chosen_n, int N, float batch,
                       float *head, const float *prob_sum, const float *count, int K, float entropy_scale, float *losses,
                       float *g, float *g_prob, void *stream) {
    if (!is_pow2(K) || K < 16 || K > 256 || N < 1) return MCQ_EUNSUPPORTED;
    if (nparts <= 0 || !num_part || !den_part || !chosen_n || !head || !prob_sum || !count || !losses || !g || !g_prob) return MCQ_EINVAL;
    hipLaunchKernelGGL(k_loss_head_tail, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), num_part, den_part, nparts,
                       chosen_n, N, batch, head, prob_sum, count, K, entropy_scale, losses, g, g_prob);
    MCQ_LAUNCH_CHECK();
    return 0;
}
int mcq_loss_head(const float *num_part, const float *den_part, long nparts, const float *chosen_n, int N, float batch,
                  float *head, void *stream) {
    if (nparts <= 0 || N <= 0 || !num_part || !den_part || !chosen_n || !head) return MCQ_EINVAL;
    hipLaunchKernelGGL(k_loss_head, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), num_part, den_part, nparts, chosen_n,
                       N, batch, head);
    MCQ_LAUNCH_CHECK();
    return 0;
}

int mcq_scales_exp(const float *centers_scale, const float *logits_scale, float speed, float *out2, void *stream) {
    if (!centers_scale || !logits_scale || !out2) return MCQ_EINVAL;
    hipLaunchKernelGGL(k_scales, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), centers_scale, logits_scale, speed, out2);
    MCQ_LAUNCH_CHECK();
    return 0;
}

// decode_backward_u8 with the trainer's epilogue: rows scaled by sa[0]*sb[0]*sc (device floats), and per-wave partials
// of <unscaled sums, dotw> in dot_part[mcq_decode_backward_waves(N, K, D)]
long mcq_decode_backward_waves(int N, int K, int D) { return (long)N * K * db_chunks(D, db_cw_of(D, K)); }

int mcq_decode_backward_u8_ex(const float *grad_out, const uint8_t *codes, long B, int N, int K, int D, float *gC,
                              const float *sa, const float *sb, float sc, const float *dotw, float *dot_part, void *stream) {
    if (K > 256) return MCQ_EUNSUPPORTED;
    if (N <= 0 || K <= 0 || D <= 0 || B < 0 || !gC) return MCQ_EINVAL;
    if (B > 0 && (!grad_out || !codes)) return MCQ_EINVAL;
    if ((dotw == nullptr) != (dot_part == nullptr)) return MCQ_EINVAL;
    // the caller sized dot_part by mcq_decode_backward_waves: the wide kernel must be the one that runs when D % 4 == 0
    if ((D & 3) == 0 && db_cw(grad_out, gC, D, K, D, 0, dotw) != db_cw_of(D, K)) return MCQ_EINVAL;     // misaligned buffers
    const int rc = launch_decode_backward<uint8_t>(grad_out, codes, B, N, K, D, gC, (long)D, 0L, N, static_cast<hipStream_t>(stream),
                                                   sa, sb, sc, dotw, dot_part);
    if (rc) return rc;
    ++g_last_launches;
    return 0;
}

long mcq_loss_bwd_waves(long B, int N, int K) {
    const int rpw = K < 64 ? 64 / K : 1;
    const long rows = B * N;
    return ((rows + (long)kLossWaves * rpw - 1) / ((long)kLossWaves * rpw)) * kLossWaves;
}

int mcq_loss_bwd_ex(const float *logits, const int64_t *idx, const float *lse, long B, int N, int K, const float *g_chosen,
                    const float *g_prob, float *grad_logits, const float *bias, float *dot_part, void *stream) {
    if (!is_pow2(K) || K < 16 || K > 256 || N < 1) return MCQ_EUNSUPPORTED;
    if (B <= 0) return MCQ_EINVAL;
    if (!logits || !idx || !lse || !g_chosen || !g_prob || !grad_logits || !bias || !dot_part) return MCQ_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int rpw = K < 64 ? 64 / K : 1;
    const long rows = B * N;
    const dim3 grid((unsigned)((rows + (long)kLossWaves * rpw - 1) / ((long)kLossWaves * rpw))), block(64 * kLossWaves);
#define MCQ_LOSS_CASE(KK)                                                                                             \
    case KK: hipLaunchKernelGGL((k_loss_bwd<KK>), grid, block, 0, st, logits, idx, lse, B, N, g_chosen, g_prob, grad_logits, bias, dot_part); break;
    switch (K) {
        MCQ_LOSS_CASE(16) MCQ_LOSS_CASE(32) MCQ_LOSS_CASE(64) MCQ_LOSS_CASE(128) MCQ_LOSS_CASE(256)
        default: return MCQ_EUNSUPPORTED;
    }
#undef MCQ_LOSS_CASE
    MCQ_LAUNCH_CHECK();
    return 0;
}

int mcq_grad_tail(const float *part_c, long n_c, const float *sa, const float *sb, float sc, const float *part_l, long n_l,
                  float speed, float *out_c, float *out_l, void *stream) {
    if (n_c < 0 || n_l < 0 || (n_c > 0 && !part_c) || (n_l > 0 && !part_l)) return MCQ_EINVAL;
    hipLaunchKernelGGL(k_grad_tail, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), part_c, n_c, sa, sb, sc, part_l, n_l,
                       speed, out_c, out_l);
    MCQ_LAUNCH_CHECK();
    return 0;
}

int mcq_last_encode_launches(void) { return g_last_launches; }

int mcq_test_select(const float *scores, int cases, int per_lane, int cnt, float *out_v, int *out_p, void *stream) {
    if (!scores || !out_v || !out_p || cases <= 0 || cnt < 1 || cnt > 64) return MCQ_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (per_lane) {
        case 1: hipLaunchKernelGGL((k_test_select<1>), dim3(cases), dim3(64), 0, st, scores, cnt, out_v, out_p); break;
        case 4: hipLaunchKernelGGL((k_test_select<4>), dim3(cases), dim3(64), 0, st, scores, cnt, out_v, out_p); break;
        case 16: hipLaunchKernelGGL((k_test_select<16>), dim3(cases), dim3(64), 0, st, scores, cnt, out_v, out_p); break;
        // (negative: the same number of keys per lane in the slot-major layout, key i of a lane at position 64 * i + lane; the third
        // layout, positions in any order, is the general form followed by a rank by position)
        case -4: hipLaunchKernelGGL((k_test_select<4, kSlotMajor>), dim3(cases), dim3(64), 0, st, scores, cnt, out_v, out_p); break;
        case -16: hipLaunchKernelGGL((k_test_select<16, kSlotMajor>), dim3(cases), dim3(64), 0, st, scores, cnt, out_v, out_p); break;
        case -1004: hipLaunchKernelGGL((k_test_select<4, kAnyOrder>), dim3(cases), dim3(64), 0, st, scores, cnt, out_v, out_p); break;
        default: return MCQ_EINVAL;
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

#ifdef MCQ_STAMPS
// debug build only (-DMCQ_STAMPS, tools/exp_stamps.py): the phase stamps of the sampled waves, [kernel][wave][slot] u64 -> host
int mcq_debug_stamps(unsigned long long *host_dst, long count, int clear) {
    const long total = (long)kStampKernels * kStampWaves * kStampSlots;
    if (count > total) count = total;
    if (hipDeviceSynchronize() != hipSuccess) return MCQ_EINVAL;
    if (host_dst && count > 0 &&
        hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_stamps), (size_t)count * 8, 0, hipMemcpyDeviceToHost) != hipSuccess) return MCQ_EINVAL;
    if (clear) {
        void *p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_stamps)) != hipSuccess || hipMemset(p, 0, (size_t)total * 8) != hipSuccess) return MCQ_EINVAL;
    }
    return (int)kStampSlots;
}
#endif

const char *mcq_profile_category_name(int category) {
    return (category >= 0 && category < CAT_COUNT) ? kCatNames[category] : nullptr;
}

int mcq_profile_encode(const float *x, long B, const void *prepared, float lscale_exp, int N, int K, int D,
                       int refine_iters, void *workspace, size_t workspace_bytes, void *stream, float *ms_out,
                       int *launches_out, int cap) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool wide = K > 256;                // (entries of more than 256-entry codebooks leave as int64: no byte form)
    const size_t need = (size_t)B * N * (wide ? 8 : 1);
    uint8_t *dummy = nullptr;                 // the codes of the profiled encodes (this entry point is a measurement tool: it allocates)
    if (!ms_out || cap <= 0) return MCQ_EINVAL;
    const hipError_t me = hipMalloc(reinterpret_cast<void **>(&dummy), need ? need : 1);
    if (me != hipSuccess) return (int)me;
    for (int i = 0; i < cap; ++i) { ms_out[i] = 0.f; if (launches_out) launches_out[i] = 0; }
    int rc = 0;
    for (int only = 0; only < CAT_COUNT && rc == 0; ++only) {      // one encode per category: see Prof
        Prof prof;
        prof.stream = st;
        prof.only = only;
        rc = run_encode(x, B, prepared, lscale_exp, N, K, D, refine_iters, wide ? nullptr : dummy,
                        wide ? reinterpret_cast<int64_t *>(dummy) : nullptr, workspace, workspace_bytes, st, &prof);
        (void)hipStreamSynchronize(st);
        for (size_t i = 0; rc == 0 && i < prof.cat.size(); ++i) {
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, prof.ev[prof.first[i]], prof.ev[prof.last[i]]);
            if (prof.cat[i] < cap) { ms_out[prof.cat[i]] += ms; if (launches_out) launches_out[prof.cat[i]] += 1; }
        }
        for (hipEvent_t e : prof.ev) (void)hipEventDestroy(e);
    }
    (void)hipFree(dummy);
    return rc == 0 ? CAT_COUNT : rc;
}

}  // extern "C"
