// kernels.h — launch interface of the hand-written gfx950 kernels (device pointers resolved).
// Semantics of every op are specified by plan.h and, executable, by tests/cpu_interp.cpp.
#pragma once
#include "plan.h"
#include <hip/hip_runtime.h>
#include <cstdlib>

namespace dmx
{

struct FastDiv // n / d for n < 2^31: magic == 0 ? n >> shift : umulhi(n, magic) >> shift
{
    unsigned magic, shift;
};
inline FastDiv make_fastdiv(unsigned d)
{
    FastDiv f;
    f.magic = 0;
    f.shift = 0;
    if (d == 0)
        d = 1;
    if ((d & (d - 1)) == 0)
    {
        while ((1u << f.shift) < d)
            ++f.shift;
        return f;
    }
    unsigned s = 0;
    while ((1ull << s) < d)
        ++s;
    f.magic = (unsigned)((((unsigned long long)1 << (31 + s)) + d - 1) / d); // exact for n < 2^31
    f.shift = s - 1;
    return f;
}

struct GemmArgs
{
    const float *X;
    i64 xBS;
    int B, P1, P0, L1, L0, Cin;
    int S1, stride1, dil1, pad1, seg0, stride0, pad0, K, Kp;
    int pro;
    const float *proStats, *proW, *proB;
    int G0;
    const float *Wt, *bias;
    const unsigned short *Wb1, *Wb2; // bf16 planes of the weight blob (same element offsets as Wt), or null (igemm_split.hip)
    int N, Np;
    int epi, act;
    float *Y;
    i64 yBS;
    int ldy;
    const float *res, *scale, *epiStats, *epiW, *epiB;
    float *rowstat;
    int NB;
    const float *table;
    float tableScale;
    int Lout, Cout;
    int trS, trOff; // EPI_TRCONV: output position j = trS*p0 + r - trOff (plan.h)
    const float *rowScale; // fp16-term kernels (GEMM_FP16X3): [M][2] = {2^s, 2^-s} per A row (launch_rowscale), else null
    unsigned short *kvPl; // EPI_KPL / EPI_VT: three bf16 planes, kvPlane elements apart (plan.h IGemm::kv)
    i64 kvPlane;
    int kvCol0, kvT, kvH, kvHs;
    i64 M;
    const float *zero; // >= 16 B of zeros, 16-byte aligned: target of out-of-range staging loads
    // launch geometry (filled by launch_igemm): 1-D grid, workgroup id -> (row tile, column tile)
    unsigned tilesM, tilesN;
    FastDiv dP0, dP1; // row index m -> (b, p1, p0) without 64-bit divisions (M < 2^31 is checked by the launcher)
    int xcdMap; // 1: XCD-aware mapping (all column tiles of a row tile on ONE XCD, adjacent in dispatch order)
};

// direct (register-resident weights, no LDS) kernels for the HBM-bound layers; -1 if the shape is
// not in their table
int launch_dgemm(const GemmArgs &a, hipStream_t s, bool dry = false);

// The frequency branch's whole DConv residual branch with the (C, T) row of one (segment, bin) resident on the CU
// (dconv_row.hip; plan.h OP_DCONV_ROW). Pointers of the two layers (dilation 1, 2) into the packed model.
struct DconvRowArgs
{
    float *x; // [B][T][F][C], in place
    int B, T, F, C, hid;
    const float *img[2]; // per layer (dilation 1, 2): the weight image in the kernel's LDS layout (plan.h DconvRowGeo)
    float eps;
    const float *zero;
    // filled by the launcher
    int rowsPerXcd;
    double n1, invN1, invN1m1, n2, invN2, invN2m1; // element counts of the two GroupNorms (hid T, 2C T) and the reciprocals of n, n - 1
};
// -1 when no kernel exists for the shape; dry = availability check only (needs B, T, F, C, hid)
int launch_dconv_row(const DconvRowArgs &a, hipStream_t s, bool dry = false);

// returns 0, or -1 when the (tile, prologue, epilogue) combination is not instantiated;
// dry = true only checks availability
int launch_igemm(int cfg, const GemmArgs &a, hipStream_t s, bool dry = false);
// linear layers on a 256x128 tile with four waves of 128x64 (igemm_lin256.hip); -1 when the op is not a plain linear layer
int launch_igemm_lin256(const GemmArgs &a, hipStream_t s, bool dry = false);
// GEMM_BF16X3 contexts: the same tiles with exact bf16 operand splits on the bf16 matrix pipe (igemm_split.hip); -1 = not available
// arith 0: bf16 terms (DMX_GEMM_BF16X3); 1: fp16 terms (DMX_GEMM_FP16X3: Wb1 = the fp16 plane, rowScale set) - exists for the
// linear-layer kernel only, returns -1 elsewhere; a dry call with arith 1 needs the op's whole geometry in `a`
int launch_igemm_split(int cfg, const GemmArgs &a, hipStream_t s, bool dry = false, int arith = 0);
// 256 / 192: that launch takes a wide tile of igemm_split_linw_kernel (128 x 256 / 128 x 192), decided per launch; 0: it does not.
// Same bits either way
int igemm_split_is_wide(int cfg, const GemmArgs &a);
// per-row scales of a linear layer's A operand (rows of K contiguous floats, the addressing of `a`): out[m] = rowscale_of(max |a|)
void launch_rowscale(const GemmArgs &a, float *out, hipStream_t s);
// the fp16 three-term split applied to an array under ONE scale 2^sexp: planes [3][n] fp16 bit patterns (unit test of the split)
void launch_split3h_debug(const float *d_x, i64 n, int sexp, unsigned short *d_planes, hipStream_t s);
// the kernels' three-term activation split applied to an array: planes [3][n] bf16 bit patterns (unit test of the split)
void launch_split3_debug(const float *d_x, i64 n, unsigned short *d_planes, hipStream_t s);

struct ReduceArgs
{
    const float *rowstat;
    float *out;
    int B, R, NB, G0;
    double count;
    int mode;
    float eps;
    double *scratch;
    int nchunk;
};
void launch_stats_reduce(const ReduceArgs &a, hipStream_t s);

struct StftArgs
{
    const float *mix;
    float *x, *rowstat, *rowstatT;
    int B, T, seg, pad;
    const float *window, *twiddle;
};
void launch_stft(const StftArgs &a, hipStream_t s);

struct LnArgs
{
    const float *x;
    float *y;
    int rows, D, rowsPerBatch;
    const float *w, *b, *pe;
    float eps;
};
void launch_layernorm(const LnArgs &a, hipStream_t s);

struct GnArgs
{
    const float *x;
    float *y;
    const float *res;
    int B, rows, C;
    const float *stats, *w, *b;
};
void launch_gn_apply(const GnArgs &a, hipStream_t s);

struct AttnArgs
{
    const float *q, *k, *v;
    float *o;
    int ldq, ldk, ldv, ldo;
    i64 qB, kB, vB, oB;
    int B, Tq, Tk, H, hs;
    float scale;
    unsigned nQt; // query tiles per (batch, head) (filled by launch_attention)
    int xcdMap;   // 1: all query tiles of one (batch, head) on ONE XCD (its K/V stay in that XCD's L2)
    // LocalState attention of Demucs v3 (launch_attention_local): decay logits of query s, head h, term n at
    // decay[b * dB + s * ldd + 4 h + n]; null for the transformer's attention
    const float *decay;
    int ldd;
    i64 dB;
    // bf16 operand planes written by the K / V projections (plan.h Attention::kpl / vt), or null
    const unsigned short *kpl, *vt;
    i64 kvPlane; // elements per plane = B * Tk * H * hs
};

void launch_attention(const AttnArgs &a, hipStream_t s);
// GEMM_BF16X3 contexts: the same attention with both products on the bf16 matrix pipe through exact three-term operand
// splits (attention_split.hip); -1 = no kernel for this head dim. dry = availability check only
int launch_attention_split(const AttnArgs &a, hipStream_t s, bool dry = false);
// Demucs v3 LocalState (src/layers.cpp:533-721) on the same flash kernel: scores + decay penalty, diagonal = -100;
// head dims 48 / 96. Returns -1 for other shapes.
int launch_attention_local(const AttnArgs &a, hipStream_t s);

struct IstftArgs
{
    const float *x, *stats;
    float *frames;
    int B, T, S;
    const float *window, *twiddle;
};
void launch_istft(const IstftArgs &a, hipStream_t s);

struct OlaArgs
{
    const float *frames, *xt, *statsT, *wss;
    float *out;
    int B, T, S, seg, pad;
};
void launch_ola(const OlaArgs &a, hipStream_t s);
// ISTFT + overlap-add + time-branch sum in one kernel (the inverse frames stay in registers)
struct IstftOlaArgs
{
    const float *x, *stats, *xt, *statsT, *wss, *window, *twiddle;
    float *out;
    int B, T, S, seg, pad;
    int nch, fpc; // frame chunks per (batch, source) and frames per chunk (filled by the launcher)
    const float *rden; // (1 / 4096) / (wss + 1e-8), same indexing as wss (plan.h Ola::rden)
};
void launch_istft_ola(const IstftOlaArgs &a, hipStream_t s);

// ---- Demucs v3 (v3.hip; plan.h OP_GROUP_STATS / OP_GN_ACT / OP_LSTM / OP_LOCAL_ATTN)
struct GroupStatsArgs
{
    const float *x;
    float *out;
    double *partials; // [B][G][32 chunks][2]
    int B, rows, C, G;
    float eps;
};
void launch_group_stats(const GroupStatsArgs &a, hipStream_t s);
struct GnActArgs
{
    const float *x;
    float *y;
    const float *stats, *res, *w, *b, *scale;
    int B, rowsIn, C, G, mode, rowOff, rowsOut;
};
void launch_gn_act(const GnActArgs &a, hipStream_t s);
struct LstmArgs
{
    const float *xproj, *whh;
    float *out;
    void *gran;       // h exchange granules, lstm_sync_floats(B, H) floats; zeroed by the launcher
    unsigned *status; // raised when a bounded spin timed out (checked by dmx_ctx_synchronize)
    int B, T, H;
};
int launch_lstm(const LstmArgs &a, hipStream_t s); // -1: unsupported hidden size; -2: the grid of spinning workgroups cannot be co-resident on this device
struct LocalAttnArgs
{
    const float *qkvd;
    float *out;
    int B, T, H, ld;
};
int launch_local_attn(const LocalAttnArgs &a, hipStream_t s); // -1: unsupported shape

// ---- track level (model_apply.cpp:60-288) ----
// partial (sum, sumsq) of the mono reference (mean over channels); audio interleaved [n][2]
void launch_track_stats(const float *audio, i64 n, double *partials, int nblk, hipStream_t s);
// stats[0]=mean, stats[1]=std (unbiased) from partials
void launch_track_stats_final(const double *partials, int nblk, i64 n, float *stats, hipStream_t s);
// chunk extraction: mixes[i] = segment `segIdx[i]` of the normalised, shifted, zero-padded track,
// centred in a zero segment (segment_inference, model_apply.cpp:250-288). segIdx is a HOST array
// (passed to the kernel by value, kMax per launch).
struct TrackSegIdx
{
    static const int kMax = 64;
    int v[kMax];
};
void launch_track_gather(const float *audio, i64 n, const float *stats, int shiftOffset, i64 seg, i64 stride,
                         i64 len, const int *segIdx, int nIdx, float *mixes, hipStream_t s);
// overlap-add of nSeg segment outputs [nSeg][S][2][seg] (segment ids 0..nSeg-1) into planes
// [planeBase, planeBase + nPlanes) (plane = stem*2 + channel) and samples [i0, i1) of out.
// layout 0: planar [S][2][n]; layout 1: Eigen column-major image (s + S*(c + 2*i))
// gBase: segOut[0] is segment gBase (a device that holds only a stretch of the segments; every segment the
// samples [i0, i1) touch must be in the buffer)
void launch_track_ola(const float *segOut, int nSeg, int S, i64 seg, i64 stride, i64 len, i64 n, int shiftOffset,
                      const float *stats, float *out, int layout, int planeBase, int nPlanes, i64 i0, i64 i1, hipStream_t s,
                      int gBase = 0);
// dst[r*dpitch + i] = src[r*spitch + i], r < rows, i < width (floats)
void launch_copy_rows(float *dst, i64 dpitch, const float *src, i64 spitch, i64 width, int rows, hipStream_t s);
// dst[i] = fp16 bit pattern of src[i], round to nearest even (the opt-in fp16 weight plane, api.cpp dmx_model_fp16_plane)
void launch_f32_to_f16(const float *src, unsigned short *dst, i64 n, hipStream_t s);
// interleaved <-> planar helpers
void launch_planar_to_interleaved(const float *src, float *dst, i64 n, hipStream_t s);


#ifdef __HIPCC__
// erf for the exact GELU 0.5 v (1 + erf(v / sqrt 2)) (/root/reference/src/layers.hpp:51-63, conv.hpp:203-204).
// Branch-free two-range minimax fit, max abs error 9.7e-8 (~1.6 ulp at 1) against scipy.special.erf over
// [-6, 6] in fp32 arithmetic (fit + check: DESIGN.md section 2.2): |x| < 0.92: x * A(x^2);
// else sign(x) * (1 - 2^B(min(|x|, 4))). 19 VALU instructions incl. one v_exp_f32; the ocml erff costs
// ~45 on a wave whose lanes straddle its two ranges, and a 128x128 GELU epilogue evaluates 64 per lane.
__device__ __forceinline__ float dmx_erff(float x)
{
    const float t = fminf(fabsf(x), 4.0f);
    const float s = x * x;
    float a = fmaf(-0.0005843715625815094f, s, 0.004958246368914843f);
    a = fmaf(a, s, -0.026736384257674217f);
    a = fmaf(a, s, 0.11280667781829834f);
    a = fmaf(a, s, -0.3761231601238251f);
    a = fmaf(a, s, 1.1283791065216064f);
    a *= x;
    float b = fmaf(0.0002812488819472492f, t, -0.004376694560050964f);
    b = fmaf(b, t, 0.03201308846473694f);
    b = fmaf(b, t, -0.149833083152771f);
    b = fmaf(b, t, -0.9193801283836365f);
    b = fmaf(b, t, -1.6268224716186523f);
    b = fmaf(b, t, -0.0002986486360896379f);
    const float r = copysignf(1.0f - __builtin_amdgcn_exp2f(b), x);
    return t < 0.92f ? a : r;
}
__device__ __forceinline__ float dmx_gelu(float v) { return 0.5f * v * (1.0f + dmx_erff(v * 0.70710678118654752440f)); }
#endif

} // namespace dmx
