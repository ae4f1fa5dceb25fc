// plan.h — the per-segment execution plan: a flat list of ops over two address spaces.
//
//   W space : packed model weights (one float blob per model, resident in HBM)
//   A space : per-context arena (constants such as window / twiddles / positional
//             tables first, then activations), one float blob per context in HBM
//
// Every `i64` field named *_w is an element offset into W, every other offset is into
// A. The plan is pure data: it is built once per context by plan.cpp (shapes are
// static for a given segment length and batch), executed by the HIP kernels in
// engine.cpp, and - in tests only - interpreted on the CPU by tests/cpu_interp.cpp to
// validate packing and index math without a GPU.
//
// Activation layouts (all fp32, channels-last; chosen so that every conv in the model
// is a GEMM whose A rows are contiguous runs of memory):
//   freq branch   [B][T][F][C]   (T = STFT frames of the segment, F = freq bins)
//   time branch   [B][L][C]
//   tokens        [B][tok][D]    (freq tokens tok = t*8 + f: exactly [T][F=8][C])
// which is also the physical order of the reference's column-major Eigen tensors
// (C fastest; /root/reference/src/tensor.hpp:24-28).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace dmx
{
typedef int64_t i64;
static inline int rup(int v, int m) { return (v + m - 1) / m * m; }

enum OpKind
{
    OP_IGEMM = 0,
    OP_STATS_REDUCE,
    OP_STFT,
    OP_LAYERNORM,
    OP_GN_APPLY,
    OP_ATTENTION,
    OP_ISTFT,
    OP_OLA,
    OP_TAP, // no-op marker: names an activation for the debug-tap API
    // ---- Demucs v3 (hdemucs_mmi) levels 4 / 5 and its decoders (plan_v3.cpp)
    OP_GROUP_STATS, // GroupNorm statistics of G channel groups straight from the tensor
    OP_GN_ACT,      // GroupNorm apply (+GELU | +GLU (+LayerScale, residual)) with an optional row crop
    OP_LSTM,        // one bidirectional LSTM layer (recurrent part; the input projection is an OP_IGEMM)
    OP_LOCAL_ATTN,  // LocalState attention core (scores + decay + softmax over keys + weighted content)
    OP_DCONV_ROW,   // the frequency branch's whole DConv residual branch, one (segment, bin) row resident per workgroup
};

enum Prologue
{
    PRO_NONE = 0,
    PRO_AFFINE = 1,  // a = (a - st[b].mean) * st[b].scale          (z-norm fused into enc0)
    PRO_GN_GELU = 2, // a = gelu((a - st[g].mean)*st[g].scale*pw[k] + pb[k])
};

enum Epilogue
{
    EPI_LINEAR = 0,            // v = acc+bias; act; += res; store [row][n]
    EPI_SCALE_RES = 1,         // v = res + (acc+bias)*scale[n]; store
    EPI_GLU = 2,               // paired cols: v = a*sigmoid(b) (+ tableScale*table[p0][c]); store [row][c]
    EPI_GN_GLU_SCALE_RES = 3,  // paired cols after GroupNorm: v = res + scale[c]*glu(gn(a),gn(b))
    EPI_STATS_ONLY = 4,        // row statistics of (acc+bias) only, nothing stored
    EPI_TRCONV = 5,            // n=(r,co): j = 4*p0 + r - 2; store [(b,p1,j)][co]; act; += res
    EPI_STATS_FACT = 6,        // row statistics of a (N x K) linear map y = W a + b WITHOUT forming y: the weights
                               // are the factor [L; u; v] (hid = Cout rows of L with L^T L = W^T W, u = W^T 1,
                               // v = W^T b; bias = [0.., sum b, sum b^2 / 2]), so with z = acc + bias:
                               //   sum_n y_n   = z[hid]          sum_n y_n^2 = sum_{k<hid} z[k]^2 + 2 z[hid+1]
    // GEMM_BF16X3 plans with PlanOpts::kvPlanes (plan.cpp, transformer): the K and V projections leave their results as the bf16
    // operand planes of the attention kernel instead of fp32 (attention_split.hip reads them straight into LDS):
    EPI_KPL = 7,               // v = acc+bias; columns < kvCol0: store fp32 [row][n]; columns >= kvCol0: three bf16 planes
                               // [plane][row][n - kvCol0] of the exact split v = v1 + v2 + v3 (igemm_common.h split3_pk)
    EPI_VT = 8,                // v = acc+bias as three bf16 planes of V^T in the attention kernel's LDS tile order:
                               // [plane][b][head][key tile][dim][64 keys, slot order] (attention_split.hip)
};

// Statistics records are 4 floats: {mean, scale, std, 0}; scale is rstd = 1/sqrt(var+eps)
// (MODE_RSTD; GroupNorm/LayerNorm, Q3 unbiased var) or 1/(std+eps) (MODE_ZNORM).
enum StatsMode
{
    MODE_RSTD = 0,
    MODE_ZNORM = 1,
};

struct IGemm
{
    // ---- rows: m -> (b, p1, p0), M = B*P1*P0
    int B, P1, P0;
    // ---- A operand: input tensor [B][L1][L0][Cin] at `x` with explicit batch stride
    i64 x, xBatchStride;
    int L1, L0, Cin;
    int S1, stride1, dil1, pad1; // taps along axis 1: in1 = p1*stride1 + s1*dil1 - pad1
    int seg0, stride0, pad0;     // inner contiguous run: elements [(p0*stride0-pad0)*Cin, +seg0)
                                 // of the inner row, zero outside [0, L0*Cin)
    int K;                       // S1*seg0 (logical K); weight rows are Kp = roundup(K,16) long
    int Kp;
    // ---- prologue
    int pro;
    i64 proStats; // A: PRO_AFFINE [B][4]; PRO_GN_GELU [B*G0][4]
    i64 proW_w, proB_w; // W: [Kp] (PRO_GN_GELU)
    int G0;       // group of a row: g = b*G0 + (G0 > 1 ? p0 : 0)
    // ---- B operand
    i64 w_w;    // W: [Np][Kp], zero padded
    i64 bias_w; // W: [Np]
    int N, Np;
    // ---- epilogue
    int epi, act;             // act: 0 none, 1 exact GELU
    i64 y, yBatchStride;      // A
    int ldy;
    i64 res;                  // A (same indexing as y) or -1
    i64 scale_w;              // W: per output channel ([Np] for SCALE_RES, [C] for GN_GLU_SCALE_RES) or -1
    i64 epiStats;             // A: [B*G0][4] for EPI_GN_GLU_SCALE_RES
    i64 epiW_w, epiB_w;       // W: [Np] GroupNorm affine in packed column order
    i64 rowstat;              // A: [M][NB][2] partial (sum, sumsq) per row and column block, or -1
    int NB;                   // number of column blocks writing rowstat (= ceil(Np/BN)); set by engine cfg
    i64 table_w;              // W: [P0][C] (EPI_GLU freq-embedding add) or -1
    float tableScale;
    int Lout, Cout;           // EPI_TRCONV: output positions per (b,p1) row, channels
    int trS, trOff;           // EPI_TRCONV: n = (r, co), r < trS; output position j = trS*p0 + r - trOff
                              // (k8/s4 with the 2-sample crop: 4, 2; v3's uncropped k8/s4: 4, 0; k4/s2: 2, 0)
    int cfg;                  // tile configuration index (engine)
    int split;                // GEMM_BF16X3 / GEMM_FP16X3 contexts (api.cpp split_kind): 1 = run on the exact-split bf16 kernel (igemm_split.hip); 2 = on its fp16-term form
    // EPI_KPL / EPI_VT: A (float offset) of three bf16 planes of B*kvT*(kvH*kvHs) elements each; rows are tokens (P1 = kvT,
    // P0 = 1), kvT a multiple of 64; -1 otherwise
    i64 kv;
    int kvCol0, kvT, kvH, kvHs;
    // GEMM_FP16X3 plans: this op (a transformer linear layer kept on the 128- / 64-row tiles at every batch size) may take the
    // fp16-term kernel; every other op of such a plan runs exactly as in a GEMM_BF16X3 plan
    int hterms = 0;
};

struct StatsReduce
{
    i64 rowstat; // A: [B][R][NB][2]
    i64 out;     // A: [B*G0][4]
    int B, R;    // R rows per batch element
    int NB;
    int G0;      // G0 > 1: row r belongs to group r % G0; else one group per batch element
    double count; // elements per group
    int mode;
    float eps;
    i64 scratch; // A: [B][nchunk][4] (two doubles) partials for the G0 <= 1 two-stage reduction, or -1
    int nchunk;
};

struct Stft
{
    i64 mix;  // A: [B][seg][2] interleaved stereo
    i64 x;    // A: [B][T][2048][4] CaC (re0,im0,re1,im1), un-normalised
    i64 rowstat;  // A: [B][T][1][2] (sum, sumsq) of the CaC frame (freq z-norm)
    i64 rowstatT; // A: [B][T][1][2] (sum, sumsq) of raw mix samples [t*1024,(t+1)*1024) x 2ch (time z-norm)
    int B, T, seg, pad;
    i64 window, twiddle; // A constants: [4096] hann, [2048][2] exp(-2 pi i k/4096)
};

struct LayerNorm
{
    i64 x, y; // A: [rows][D]
    int rows, D, rowsPerBatch;
    i64 w_w, b_w; // W
    i64 pe;       // A constant [rowsPerBatch][D] added after the affine, or -1
    float eps;
};

struct GnApply
{
    // v = x; if stats >= 0: v = (v - mean_b)*rstd_b*w[c] + b[c]; if res >= 0: v += res; y = v
    i64 x, y, res; // A: [B][rows][C]
    int B, rows, C;
    i64 stats; // A: [B][4] or -1
    i64 w_w, b_w;
};

struct Attention
{
    i64 q, k, v, o; // A
    int ldq, ldk, ldv, ldo;
    i64 qBatch, kBatch, vBatch, oBatch; // batch strides (elements)
    int B, Tq, Tk, H, hs;
    float scale; // 1/sqrt(hs)
    int split;   // GEMM_BF16X3 contexts (api.cpp): run on the exact-split bf16 kernel (attention_split.hip)
    i64 kpl = -1, vt = -1; // A: bf16 operand planes of K ([3][B][Tk][H*hs]) and V^T ([3][B][H][Tk/64][hs][64]) written by the
                           // projections (EPI_KPL / EPI_VT), or -1: the kernel splits fp32 k / v itself
};

struct Istft
{
    i64 x;      // A: [B][T][2048][4S] decoder output (normalised domain)
    i64 stats;  // A: [B][4] z-norm record of the freq branch (mean, scale, std)
    i64 frames; // A: [B][S][2][T][4096] windowed inverse frames (y * hann)
    int B, T, S;
    i64 window, twiddle;
    int fused; // 1: the GPU runs this op inside the following OP_OLA (fft.hip istft_ola_kernel: the frames never
               // reach HBM); the CPU interpreter keeps the two-step form, which defines the arithmetic
};

struct Ola
{
    i64 frames;  // as Istft
    i64 xt;      // A: [B][seg][2S] time-branch decoder output (normalised domain)
    i64 statsT;  // A: [B][4] z-norm record of the time branch
    i64 wss;     // A constant: [(T+4-1)*1024 + 4096] window sum-square of T+4 frames
    i64 out;     // A: [B][S][2][seg] planar
    int B, T, S, seg, pad;
    i64 x, stats, window, twiddle; // operands of the preceding OP_ISTFT (fused execution), -1 otherwise
    i64 rden;    // A constant, same indexing as wss: (1 / 4096) / (wss + 1e-8) - the fused GPU kernel multiplies by it instead of
                 // dividing twice per term (<= 1 ulp per term from the reference's y / 4096 / (wss + 1e-8), dsp.cpp:151-185)
};

// ---- Demucs v3 ops. Tensors are [B][rows][C] channels-last like everything else.
// GroupNorm statistics (UNBIASED variance, Q3) of G groups: group g = channels [g*C/G, (g+1)*C/G) over ALL rows of a
// batch element (/root/reference/src/layers.hpp:125-168; rows = T, or T*8 for decoder.1's norm2).
struct GroupStats
{
    i64 x;   // A: [B][rows][C]
    i64 out; // A: [B][G][4] {mean, rstd, std, 0}
    int B, rows, C, G;
    float eps;
    i64 scratch; // A: [B][G][32][2] doubles (chunk partials of the two-launch reduction; 8-byte aligned)
};
// y[b][r][c'] = f(x[b][r + rowOff][.]) for r < rowsOut:
//   mode 0: v = gn(x[c]);  mode 1: v = gelu(gn(x[c]));  mode 2 (GLU): v = gn(x[c]) * sigmoid(gn(x[c + C/2])), C' = C/2;
//   then if scale_w >= 0: v *= scale[c'];  if res >= 0: v += res[b][r][c'].
//   gn(x[c]) = (x - mean_g) * rstd_g * w[c] + b[c], g = c / (C/G).
struct GnAct
{
    i64 x, y, stats, res; // A (res may alias y)
    i64 w_w, b_w, scale_w; // W ([C], [C], [C/2] or -1)
    int B, rowsIn, C, G, mode, rowOff, rowsOut;
};
// One bidirectional LSTM layer, zero initial state (/root/reference/src/lstm.cpp:68-147). The input projection
// xproj = x W_ih^T + (b_ih + b_hh) of BOTH directions comes from an OP_IGEMM: [B][T][2][4H] with the gate rows of a
// direction in (unit, gate) order: column 4*j + g, g = i|f|g|o. whh: [2][4H][H] in the same row order.
// out[b][t][dir*H + j] = h_t of direction dir (forward | backward concatenated, lstm.cpp:136-143).
struct Lstm
{
    i64 xproj; // A: [B][T][8H]
    i64 whh_w; // W: [2][4H][H]
    i64 out;   // A: [B][T][2H]
    i64 sync;  // A: scratch of the cooperative kernel (h exchange granules), zeroed by the launcher; >= lstm_sync_floats(B, H) floats
    int B, T, H;
};
i64 lstm_sync_floats(int B, int H);
i64 lstm_xchg_floats(int B, int T, int H); // >= lstm_sync_floats: the 4-byte exchange image [groups][T][H][16] (experiment)
// LocalState attention core (/root/reference/src/layers.cpp:533-721), heads = 4, 4 decay rates.
// qkvd: [B][T][ld] = [query H | key H | content H | decay logits 16] per position (one OP_IGEMM).
// dots(t, s) = q_s . k_t / sqrt(H/4) - sum_n (n+1) |t-s| / 2 * sigmoid(d[s][4h+n]) / 2, diagonal = -100;
// softmax over the KEY index t per query s; out[s] = sum_t w(t, s) content[t].
struct LocalAttn
{
    i64 qkvd; // A
    i64 out;  // A: [B][T][H]
    int B, T, H, ld;
};

// DConv residual branch of the FREQUENCY branch, both layers, in place on x [B][T][F][C]
// (/root/reference/src/layers.cpp:152-375 with the bins as the batch, src/encdec.cpp:43-45,203-207). Per row (b, f), with
// X = x[b][.][f][.] (T x C), for layer j = 0, 1 (dilation d = 1, 2):
//   h  = Conv1d(C -> hid, k3, dilation d, padding d)(X)            weights k1.Wt [16][3C] (k = tap*C + c), k1.b
//   hn = gelu(GroupNorm(1, hid)(h))                                statistics over hid x T, unbiased variance (Q3); gn1.w / gn1.b
//   y  = Conv1d(hid -> 2C, 1x1)(hn)                                k2.Wt [2C][16] rows in paired order (model_pack.cpp), k2.b
//   X += scale * glu(GroupNorm(1, 2C)(y))                          statistics over 2C x T; gn2.w / gn2.b (paired order), scale [C]
// The GPU kernel (dconv_row.hip) takes the 2C-wide statistics from the factor k2f.Wt / k2f.b (see EPI_STATS_FACT) and keeps
// the row in registers; this op replaces the ten-op chain K1 / r1 / K2 / r2 / K3 per layer of Builder::dconv where
// dconv_row_lds_bytes() says a kernel exists.
struct DconvRow
{
    i64 x;
    int B, T, F, C, hid;
    i64 k1_w[2], k1_b[2], gn1_w[2], gn1_b[2], k2_w[2], k2_b[2], k2f_w[2], k2f_b[2], gn2_w[2], gn2_b[2], scale_w[2]; // W
    i64 img_w[2]; // W: the same weights of a layer as ONE image in the row kernel's LDS layout (model_pack.cpp "rowimg")
    float eps;
};
// Layout of the row kernel's per-layer weight image and LDS footprint (must match dconv_row.hip RowGeo; its launcher checks).
// image = [K1 side: Wp[NP][C + 8]: row 16 (q / 4) + 4 h + q % 4 for slot q = o RPL + c (o = 0 centre tap, 1 tap 0, 2 tap 2; NP = 16
// ceil(3 RPL / 4)) = k1.Wt[RPL h + c][tap * C + .], unused rows zero | K3 side: W3 planes [RPL][2C][4] =
// k2.Wt[row][RPL h + c], factor planes [RPL][16][4] = k2f.Wt[n][RPL h + c], constants k2.b | gn2.w | gn2.b (2C each) | scale (C)
// | k1.b | gn1.w | gn1.b | k2f.b (16 each)], each side padded to whole 256-float pieces.
struct DconvRowGeo
{
    int HP, RPL, NP, WS, nWp, oW3, oLf, oCst, nWk;
    bool resident; // both layers' K1 sides stay in LDS (C = 48)
};
bool dconv_row_geo(int C, int hid, DconvRowGeo &g); // false: no row kernel for (C, hidden width)
i64 dconv_row_image_floats(int C, int hid);        // nWp + nWk, 0 = no kernel
// LDS bytes of a workgroup for (C, hidden width, T); 0 = no kernel for the shape
size_t dconv_row_lds_bytes(int C, int hid, int T);

struct Tap
{
    i64 off;
    int shape[4]; // physical dims (batch excluded), unused = 0
    i64 batchStride;
};

struct Op
{
    int kind;
    int stream; // 0 = freq branch / main, 1 = time branch (the engine runs them on two HIP streams)
    int waitOp = -1;     // index of the latest op of the OTHER stream this op has a data hazard with
                         // (RAW, WAR or WAW on the arena), -1 if none or already implied (compute_deps)
    bool signals = false; // some later op of the other stream waits on this op
    std::string name;
    IGemm g;
    StatsReduce sr;
    Stft stft;
    LayerNorm ln;
    GnApply gn;
    Attention at;
    Istft istft;
    Ola ola;
    Tap tap;
    GroupStats gs;
    GnAct ga;
    Lstm lstm;
    LocalAttn la;
    DconvRow dr;
};

// geometry of one segment (model.hpp:19-24,618-625 generalised to any length)
struct Geo
{
    i64 seg, le, pad, pad_end, padded, nfr;
    i64 Lt[5];
};
Geo make_geo(i64 seg);

struct PackedModel
{
    int arch = 4; // 4: HTDemucs v4 (dmc4 / dmc6); 3: Demucs v3 hdemucs_mmi (dmc3)
    int n_sources = 4;
    int dim = 512;
    int n_tensors = 0;
    std::vector<float> blob;                       // W space
    std::vector<std::pair<std::string, i64>> index; // packed array name -> offset
    i64 find(const std::string &name) const;
    bool has(const std::string &name) const;
};

// tile configurations of the igemm kernel (engine.cpp instantiates exactly these)
struct TileCfg
{
    int BM, BN;
};
static const TileCfg kTileCfgs[] = {
    {128, 128}, // 0
    {64, 64},   // 1
    {128, 96},  // 2
    {128, 48},  // 3
    {256, 16},  // 4
    {128, 32},  // 5
    {128, 64},  // 6
    {64, 128},  // 7  same column decomposition as 0 (2 waves x 4 fragments): bit-identical row statistics
    {16, 256},  // 8  kDirectCfg: dgemm.hip direct kernels, one wave owns 16 rows x all columns (NB = 1)
    // half-height siblings for launches that would leave CUs idle (few segments in flight). Same column
    // decomposition as the tile they replace (waves x fragments along N), so the row statistics - and every
    // other output bit - are identical; 64x64 (2 waves x 2 fragments along N) only replaces ops without row statistics.
    {64, 64},   // 9   of 7 (ops without row statistics only)
    {64, 96},   // 10  of 2
    {64, 48},   // 11  of 3
    {64, 32},   // 12  of 5
    {64, 64},   // 13  of 6 (4 x 1 waves)
    {128, 16},  // 14  of 4
    // quarter-height siblings for one or two segments per call (M = 1344 / 2688 rows: even the half-height tiles
    // leave most CUs with a single workgroup)
    {32, 128},  // 15  of 7 (2 x 2 waves, 1 x 4 fragments: the column decomposition of 0 and 7, row statistics identical)
    {32, 64},   // 16  of 9
    {256, 128}, // 17  double-height sibling of 0 (8 waves, one workgroup per CU); experiment behind DMX_TALL=1
    {256, 128}, // 18  the same tile with FOUR waves of 128x64 and 16-deep K-tiles (two workgroups per CU); experiment, DMX_TALL=2
    {256, 128}, // 19  igemm_lin256.hip: that tile with its own pipelined loop, linear layers only
                //     (same column decomposition as 0 / 7, bit-identical results)
    {256, 96},  // 20  four waves of 64x96, 16-deep K-tiles, plain loop: the short-K ops of the 128x96 family (plan.cpp); all of
                //     that family with DMX_TALL=3 (probe)
};
static const int kNumTileCfgs = 21;
static const int kDirectCfg = 8;
// cfg -> its half-height sibling (-1: none); stat = the op writes row statistics
int half_cfg(int cfg, bool stat);
// tile for an op at the ACTUAL number of rows M (all segments in flight): the largest tile of the chain
// cfg -> half_cfg(cfg) -> ... that keeps the 256 CUs evenly busy
int refine_cfg(int cfg, i64 M, int N, bool stat);
// shapes served by the direct kernels (must match the table in dgemm.hip)
bool direct_available(int N, int S1, int seg0, int pro, int epi);
// M1 = rows per batch element: the choice never depends on the batch size, so results are
// bit-identical for any batching / sharding of segments
int choose_cfg(i64 M1, int N, bool paired);

// how the MFMA-bound convs / linears (and, where built, attention) form their fp32 products
enum GemmMode
{
    GEMM_F32 = 0,    // v_mfma_f32_16x16x4_f32: fp32 operands, one k-ordered fmaf chain per output
    GEMM_BF16X3 = 1, // exact operand splits a = a1 + a2 + a3, w = w1 + w2 (bf16 terms) on the bf16 matrix pipe, fp32 accumulate
    GEMM_FP16X3 = 2, // opt-in: as GEMM_BF16X3 (same plan), except that the linear layers run with fp16 terms under a per-row
                     // power-of-two scale where their weights allow (api.cpp split_kind): bounded, not exact
};
struct PlanOpts
{
    int gemm = GEMM_F32;
    int kvPlanes = 0; // GEMM_BF16X3 only: K / V projections write the attention kernel's bf16 operand planes (EPI_KPL / EPI_VT)
};

struct Plan
{
    int B = 1;
    int gemm = GEMM_F32; // PlanOpts::gemm the plan was built for (tile rules differ: plan.cpp finish())
    Geo geo;
    int S = 4, D = 512;
    i64 arenaFloats = 0;
    std::vector<float> constants; // first `constants.size()` floats of A
    std::vector<Op> ops;
    i64 zeroOff = 0; // A: 64 floats that stay zero (igemm staging target for padding)
    i64 mixOff = 0; // A: [B][seg][2]
    i64 outOff = 0; // A: [B][S][2][seg]
};

// model_pack.cpp
bool load_and_pack(const std::string &path, PackedModel &pm, std::string &err);
// plan.cpp (v4) / plan_v3.cpp (v3, dispatched from build_plan on pm.arch)
void build_plan(const PackedModel &pm, i64 seg, int B, Plan &plan, const PlanOpts &opts = PlanOpts());
void build_plan_v3(const PackedModel &pm, i64 seg, int B, Plan &plan, const PlanOpts &opts = PlanOpts());
// arena ranges [lo, hi) an op reads / writes (conservative hulls); constants and W space excluded
struct Range
{
    i64 lo, hi;
};
void op_access(const Op &op, std::vector<Range> &reads, std::vector<Range> &writes);
// fills Op::waitOp / Op::signals from the access ranges (called by build_plan)
void compute_deps(Plan &plan);

} // namespace dmx
