// model_pack.cpp — dmc4 / dmc6 / dmc3 weight-file reader and repacking into GEMM-ready layouts.
//
// File format (kept surface): /root/reference/src/model_load.cpp:79-147 (reader) and
// /root/reference/scripts/convert-pth-to-ggml.py:111-140 (writer). Error behaviour
// mirrors the reference loader: open failure, bad magic, unknown tensor name or
// element-count mismatch => false + message (model_load.cpp:64-69,97-102,1065-1070,
// 1096-1105). Unlike the reference we ALSO fail when a tensor is missing.
// Demucs v3 ("dmc3", hdemucs_mmi): reader /root/reference/src/model_load.cpp:1302-2166, same container.
//
// Packing (done once on the host, then the blob lives in HBM):
//   * every conv / linear weight becomes a row-major matrix Wt[Np][Kp] (Np = N rounded
//     up to 16, Kp = K rounded up to 16, zero padded) with K ordered exactly like the
//     contiguous activation runs of the channels-last layout (tap-major, channel
//     fastest), so the conv is one GEMM with no im2col (plan.h);
//   * GLU producers get their 2C rows interleaved in blocks of 16 (a16|b16) so the two
//     gate halves of a channel sit in adjacent MFMA fragments of one wave;
//   * transposed convs k8/s4 become N = 4*Cout, K = 2*Cin: out[4q+r] = [x[q-1], x[q]] .
//     [W[:,co,r+4]; W[:,co,r]] (no zero stuffing, cf. src/conv.hpp:264-325).
#include "plan.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>

namespace dmx
{

static float half_to_float(uint16_t h)
{
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1f, m = h & 0x3ffu, f;
    if (e == 0)
    {
        if (m == 0)
            f = sign;
        else
        {
            int sh = 0;
            while ((m & 0x400u) == 0)
            {
                m <<= 1;
                ++sh;
            }
            m &= 0x3ffu;
            f = sign | ((uint32_t)(113 - sh) << 23) | (m << 13);
        }
    }
    else if (e == 31)
        f = sign | 0x7f800000u | (m << 13);
    else
        f = sign | ((e + 112) << 23) | (m << 13);
    float out;
    std::memcpy(&out, &f, 4);
    return out;
}

struct Raw
{
    std::vector<int> shape;
    std::vector<float> d;
    i64 numel() const { return (i64)d.size(); }
};


i64 PackedModel::find(const std::string &name) const
{
    for (auto &kv : index)
        if (kv.first == name)
            return kv.second;
    fprintf(stderr, "[dmx] internal error: packed array '%s' not found\n", name.c_str());
    abort();
}

bool PackedModel::has(const std::string &name) const
{
    for (auto &kv : index)
        if (kv.first == name)
            return true;
    return false;
}

namespace
{
struct Packer
{
    PackedModel &pm;
    const std::map<std::string, Raw> &raw;
    explicit Packer(PackedModel &p, const std::map<std::string, Raw> &r) : pm(p), raw(r) {}

    const Raw &get(const std::string &n) const { return raw.at(n); }

    float *alloc(const std::string &name, i64 n)
    {
        i64 off = (i64)pm.blob.size();
        off = (off + 63) / 64 * 64; // 256-byte aligned arrays
        pm.blob.resize((size_t)(off + n), 0.0f);
        pm.index.emplace_back(name, off);
        return pm.blob.data() + off;
    }
    // packed row index of logical GLU row (c, half) -- blocks of 16: a16 | b16
    static int paired_row(int c, int half) { return (c / 16) * 32 + half * 16 + (c % 16); }

    // plain vector copy, padded
    void vec(const std::string &dst, const std::string &src, int npad)
    {
        const Raw &r = get(src);
        float *p = alloc(dst, npad);
        for (i64 i = 0; i < r.numel(); ++i)
            p[i] = r.d[(size_t)i];
    }
    // vector of length 2C in paired order
    void vec_paired(const std::string &dst, const std::string &src)
    {
        const Raw &r = get(src);
        int C = (int)r.numel() / 2;
        float *p = alloc(dst, 2 * C);
        for (int c = 0; c < C; ++c)
        {
            p[paired_row(c, 0)] = r.d[(size_t)c];
            p[paired_row(c, 1)] = r.d[(size_t)(C + c)];
        }
    }
    // generic conv / linear weight (N, Cin, taps...) -> Wt[Np][Kp], k = tapidx*Cin + ci.
    // `tapmap[t]` gives the source flat tap index (over the trailing dims) of packed tap t.
    void conv(const std::string &dst, const std::string &src, int N, int Cin, const std::vector<int> &tapmap,
              bool paired, int Npad_extra = 0)
    {
        const Raw &r = get(src);
        int taps = (int)tapmap.size();
        int K = taps * Cin, Kp = rup(K, 16), Np = rup(std::max(N, Npad_extra), 16);
        float *p = alloc(dst, (i64)Np * Kp);
        int srcTaps = (int)(r.numel() / ((i64)N * Cin));
        for (int n = 0; n < N; ++n)
        {
            int row = n;
            if (paired)
            {
                int C = N / 2;
                row = (n < C) ? paired_row(n, 0) : paired_row(n - C, 1);
            }
            for (int ci = 0; ci < Cin; ++ci)
                for (int t = 0; t < taps; ++t)
                    p[(i64)row * Kp + t * Cin + ci] = r.d[(size_t)(((i64)n * Cin + ci) * srcTaps + tapmap[(size_t)t])];
        }
    }
    // transposed conv (Cin, Cout, kt), stride kt/2 (k8/s4, or v3's k4/s2) -> Wt[S*Cout][2*Cin], bias[S*Cout], S = kt/2:
    // out[S q + r] = [x[q-1], x[q]] . [W[:, co, r + S]; W[:, co, r]]
    void conv_tr(const std::string &dstW, const std::string &dstB, const std::string &srcW, const std::string &srcB, int kt = 8)
    {
        const Raw &r = get(srcW);
        const Raw &b = get(srcB);
        const int S = kt / 2;
        int Cout = (int)b.numel();
        int Cin = (int)(r.numel() / ((i64)Cout * kt));
        int N = S * Cout, K = 2 * Cin, Kp = rup(K, 16), Np = rup(N, 16);
        float *p = alloc(dstW, (i64)Np * Kp);
        for (int rr = 0; rr < S; ++rr)
            for (int co = 0; co < Cout; ++co)
                for (int ci = 0; ci < Cin; ++ci)
                {
                    i64 row = (i64)(rr * Cout + co) * Kp;
                    p[row + 0 * Cin + ci] = r.d[(size_t)(((i64)ci * Cout + co) * kt + rr + S)]; // x[q-1] * W[.., r+S]
                    p[row + 1 * Cin + ci] = r.d[(size_t)(((i64)ci * Cout + co) * kt + rr)];     // x[q]   * W[.., r]
                }
        float *pb = alloc(dstB, Np);
        for (int rr = 0; rr < S; ++rr)
            for (int co = 0; co < Cout; ++co)
                pb[rr * Cout + co] = b.d[(size_t)co];
    }

    // DConv of levels 4 / 5 of Demucs v3 (LSTM + LocalState); /root/reference/src/layers.cpp:877-1113.
    void dconv_lstm(const std::string &p, int C)
    {
        const int H = C / 4;
        for (int j = 0; j < 2; ++j)
        {
            std::string s = p + ".dconv.layers." + std::to_string(j) + ".";
            std::string d = p + ".dconv." + std::to_string(j) + ".";
            conv(d + "k1.Wt", s + "0.weight", H, C, {0, 1, 2}, false);
            vec(d + "k1.b", s + "0.bias", rup(H, 16));
            vec(d + "gn1.w", s + "1.weight", H);
            vec(d + "gn1.b", s + "1.bias", H);
            for (int layer = 0; layer < 2; ++layer)
            {
                const int In = layer == 0 ? H : 2 * H;
                // input projection of both directions: rows dir*4H + 4*unit + gate (gate = i|f|g|o, lstm.cpp:109-117),
                // bias = b_ih + b_hh (lstm.cpp:96-107)
                // (alloc() may move the blob: fill local arrays, then copy)
                std::vector<float> w((size_t)8 * H * In), bb((size_t)8 * H), u((size_t)8 * H * H);
                for (int dir = 0; dir < 2; ++dir)
                {
                    std::string sfx = "l" + std::to_string(layer) + (dir ? "_reverse" : "");
                    const Raw &wih = get(s + "3.lstm.weight_ih_" + sfx), &whh = get(s + "3.lstm.weight_hh_" + sfx);
                    const Raw &bih = get(s + "3.lstm.bias_ih_" + sfx), &bhh = get(s + "3.lstm.bias_hh_" + sfx);
                    for (int unit = 0; unit < H; ++unit)
                        for (int g = 0; g < 4; ++g)
                        {
                            const i64 dst = (i64)dir * 4 * H + 4 * unit + g, src = (i64)g * H + unit;
                            for (int k = 0; k < In; ++k)
                                w[(size_t)(dst * In + k)] = wih.d[(size_t)(src * In + k)];
                            for (int k = 0; k < H; ++k)
                                u[(size_t)(dst * H + k)] = whh.d[(size_t)(src * H + k)];
                            bb[(size_t)dst] = bih.d[(size_t)src] + bhh.d[(size_t)src];
                        }
                }
                const std::string ln = d + "lstm" + std::to_string(layer);
                std::copy(w.begin(), w.end(), alloc(ln + ".ih.Wt", (i64)w.size()));
                std::copy(bb.begin(), bb.end(), alloc(ln + ".ih.b", (i64)bb.size()));
                std::copy(u.begin(), u.end(), alloc(ln + ".hh", (i64)u.size()));
            }
            conv(d + "lin.Wt", s + "3.linear.weight", H, 2 * H, {0}, false);
            vec(d + "lin.b", s + "3.linear.bias", rup(H, 16));
            // LocalState projections as ONE GEMM: rows [query H | key H | content H | query_decay 16]
            {
                const int N = 3 * H + 16, Np = rup(N, 16);
                std::vector<float> w((size_t)Np * H, 0.f), bb((size_t)Np, 0.f);
                const char *nm[4] = {"query", "key", "content", "query_decay"};
                int row = 0;
                for (int q = 0; q < 4; ++q)
                {
                    const Raw &wq = get(s + "4." + nm[q] + ".weight"), &bq = get(s + "4." + nm[q] + ".bias");
                    const int n = (int)bq.numel();
                    for (int i = 0; i < n; ++i, ++row)
                    {
                        for (int k = 0; k < H; ++k)
                            w[(size_t)((i64)row * H + k)] = wq.d[(size_t)((i64)i * H + k)];
                        bb[(size_t)row] = bq.d[(size_t)i];
                    }
                }
                std::copy(w.begin(), w.end(), alloc(d + "qkvd.Wt", (i64)w.size()));
                std::copy(bb.begin(), bb.end(), alloc(d + "qkvd.b", (i64)bb.size()));
            }
            conv(d + "proj.Wt", s + "4.proj.weight", H, H, {0}, false);
            vec(d + "proj.b", s + "4.proj.bias", rup(H, 16));
            conv(d + "k2.Wt", s + "5.weight", 2 * C, H, {0}, false);
            vec(d + "k2.b", s + "5.bias", 2 * C);
            vec(d + "gn2.w", s + "6.weight", 2 * C);
            vec(d + "gn2.b", s + "6.bias", 2 * C);
            vec(d + "scale", s + "8.scale", C);
        }
    }

    // compress: hidden width = C / compress (8: HTDemucs v4, 4: Demucs v3)
    void dconv(const std::string &p, int C, int compress = 8, bool rowimg = false)
    {
        int C8 = C / compress, C8p = rup(C8, 4);
        for (int j = 0; j < 2; ++j)
        {
            std::string s = p + ".dconv.layers." + std::to_string(j) + ".";
            std::string d = p + ".dconv." + std::to_string(j) + ".";
            // k1: Conv1d(C -> C/8, k3, dilation): N = C8p (zero rows for the pad columns)
            conv(d + "k1.Wt", s + "0.weight", C8, C, {0, 1, 2}, false, C8p);
            vec(d + "k1.b", s + "0.bias", rup(C8p, 16));
            // GroupNorm(1, C/8) affine, indexed by k of the next GEMM
            vec(d + "gn1.w", s + "1.weight", rup(C8p, 16));
            vec(d + "gn1.b", s + "1.bias", rup(C8p, 16));
            // k2: Conv1d(C/8 -> 2C, 1x1), paired rows, K = C8p
            {
                const Raw &r = get(s + "3.weight");
                int N = 2 * C, Kp = rup(C8p, 16);
                float *w = alloc(d + "k2.Wt", (i64)N * Kp);
                for (int n = 0; n < N; ++n)
                {
                    int row = (n < C) ? paired_row(n, 0) : paired_row(n - C, 1);
                    for (int k = 0; k < C8; ++k)
                        w[(i64)row * Kp + k] = r.d[(size_t)((i64)n * C8 + k)];
                }
            }
            vec_paired(d + "k2.b", s + "3.bias");
            // k2f: statistics-only surrogate of k2 (EPI_STATS_FACT). GroupNorm(1, 2C) needs sum(y) and
            // sum(y^2) of y = W h + b per row; sum(y^2) = h^T (W^T W) h + 2 (W^T b).h + sum(b^2), so a factor
            // L with L^T L = W^T W (from the symmetric eigen-decomposition, exact for rank-deficient W too)
            // turns the 2C-column product into a (C/8 + 2)-column one. Everything in double, stored fp32.
            {
                const Raw &r = get(s + "3.weight");
                const Raw &rb = get(s + "3.bias");
                const int N = 2 * C, Kp = rup(C8p, 16), Nf = C8p + 2, Nfp = rup(Nf, 16);
                std::vector<double> Am((size_t)C8 * C8, 0.0), u((size_t)C8, 0.0), v((size_t)C8, 0.0);
                double sb = 0.0, sb2 = 0.0;
                for (int n = 0; n < N; ++n)
                {
                    const double bn = rb.d[(size_t)n];
                    sb += bn;
                    sb2 += bn * bn;
                    for (int k = 0; k < C8; ++k)
                    {
                        const double wk = r.d[(size_t)((i64)n * C8 + k)];
                        u[(size_t)k] += wk;
                        v[(size_t)k] += wk * bn;
                        for (int l = 0; l < C8; ++l)
                            Am[(size_t)k * C8 + l] += wk * (double)r.d[(size_t)((i64)n * C8 + l)];
                    }
                }
                // cyclic Jacobi: Am -> diag(lambda), Q accumulates the rotations (columns = eigenvectors)
                std::vector<double> Q((size_t)C8 * C8, 0.0);
                for (int k = 0; k < C8; ++k)
                    Q[(size_t)k * C8 + k] = 1.0;
                for (int sweep = 0; sweep < 60; ++sweep)
                {
                    double off = 0.0;
                    for (int a = 0; a < C8; ++a)
                        for (int b2 = a + 1; b2 < C8; ++b2)
                            off += Am[(size_t)a * C8 + b2] * Am[(size_t)a * C8 + b2];
                    if (off < 1e-30)
                        break;
                    for (int a = 0; a < C8; ++a)
                        for (int b2 = a + 1; b2 < C8; ++b2)
                        {
                            const double apq = Am[(size_t)a * C8 + b2];
                            if (std::fabs(apq) < 1e-300)
                                continue;
                            const double theta = (Am[(size_t)b2 * C8 + b2] - Am[(size_t)a * C8 + a]) / (2.0 * apq);
                            const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                            const double c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c;
                            for (int k = 0; k < C8; ++k)
                            {
                                const double akp = Am[(size_t)k * C8 + a], akq = Am[(size_t)k * C8 + b2];
                                Am[(size_t)k * C8 + a] = c * akp - sn * akq;
                                Am[(size_t)k * C8 + b2] = sn * akp + c * akq;
                            }
                            for (int k = 0; k < C8; ++k)
                            {
                                const double apk = Am[(size_t)a * C8 + k], aqk = Am[(size_t)b2 * C8 + k];
                                Am[(size_t)a * C8 + k] = c * apk - sn * aqk;
                                Am[(size_t)b2 * C8 + k] = sn * apk + c * aqk;
                            }
                            for (int k = 0; k < C8; ++k)
                            {
                                const double qkp = Q[(size_t)k * C8 + a], qkq = Q[(size_t)k * C8 + b2];
                                Q[(size_t)k * C8 + a] = c * qkp - sn * qkq;
                                Q[(size_t)k * C8 + b2] = sn * qkp + c * qkq;
                            }
                        }
                }
                float *wf = alloc(d + "k2f.Wt", (i64)Nfp * Kp);
                float *bf = alloc(d + "k2f.b", Nfp);
                for (int k = 0; k < C8; ++k) // row k of L = sqrt(lambda_k) * (eigenvector k)^T
                {
                    const double lam = std::max(Am[(size_t)k * C8 + k], 0.0), sq = std::sqrt(lam);
                    for (int l = 0; l < C8; ++l)
                        wf[(i64)k * Kp + l] = (float)(sq * Q[(size_t)l * C8 + k]);
                }
                for (int l = 0; l < C8; ++l)
                {
                    wf[(i64)C8p * Kp + l] = (float)u[(size_t)l];
                    wf[(i64)(C8p + 1) * Kp + l] = (float)v[(size_t)l];
                }
                bf[C8p] = (float)sb;
                bf[C8p + 1] = (float)(0.5 * sb2);
            }
            vec_paired(d + "gn2.w", s + "4.weight");
            vec_paired(d + "gn2.b", s + "4.bias");
            vec(d + "scale", s + "6.scale", C);
            // the same layer as ONE image in the LDS layout of the row-resident kernel (dconv_row.hip; layout: plan.h
            // DconvRowGeo), for the frequency branch's levels it serves: staging is then a flat copy (LDS-DMA)
            DconvRowGeo g;
            if (rowimg && dconv_row_geo(C, C8, g))
            {
                std::vector<float> img((size_t)(g.nWp + g.nWk), 0.f);
                auto at = [&](const std::string &n) { return pm.blob.data() + pm.find(n); };
                const float *k1 = at(d + "k1.Wt"), *k2 = at(d + "k2.Wt"), *k2f = at(d + "k2f.Wt");
                static const int tapOf[3] = {1, 0, 2}; // slot group o -> tap (the centre tap first)
                for (int o = 0; o < 3; ++o)
                    for (int j = 0; j < C8; ++j)
                    {
                        const int hh = j / g.RPL, c = j % g.RPL, q = o * g.RPL + c, row = 16 * (q / 4) + 4 * hh + q % 4;
                        for (int k = 0; k < C; ++k)
                            img[(size_t)(row * g.WS + k)] = k1[(i64)j * 3 * C + tapOf[o] * C + k];
                    }
                float *wk = img.data() + g.nWp;
                for (int c = 0; c < g.RPL; ++c)
                    for (int hh = 0; hh < 4; ++hh)
                    {
                        for (int r = 0; r < 2 * C; ++r)
                            wk[g.oW3 + (c * 2 * C + r) * 4 + hh] = k2[(i64)r * 16 + g.RPL * hh + c];
                        for (int r = 0; r < 16; ++r)
                            wk[g.oLf + (c * 16 + r) * 4 + hh] = k2f[(i64)r * 16 + g.RPL * hh + c];
                    }
                float *cst = wk + g.oCst;
                std::copy(at(d + "k2.b"), at(d + "k2.b") + 2 * C, cst);
                std::copy(at(d + "gn2.w"), at(d + "gn2.w") + 2 * C, cst + 2 * C);
                std::copy(at(d + "gn2.b"), at(d + "gn2.b") + 2 * C, cst + 4 * C);
                std::copy(at(d + "scale"), at(d + "scale") + C, cst + 6 * C);
                std::copy(at(d + "k1.b"), at(d + "k1.b") + 16, cst + 7 * C);
                std::copy(at(d + "gn1.w"), at(d + "gn1.w") + 16, cst + 7 * C + 16);
                std::copy(at(d + "gn1.b"), at(d + "gn1.b") + 16, cst + 7 * C + 32);
                std::copy(at(d + "k2f.b"), at(d + "k2f.b") + 16, cst + 7 * C + 48);
                std::copy(img.begin(), img.end(), alloc(d + "rowimg", (i64)img.size())); // (alloc may move the blob: copied last)
            }
        }
    }
};
} // namespace

static bool expected_shapes(int ns, std::map<std::string, i64> &exp)
{
    // element counts per tensor name; /root/reference/src/model.hpp:26-554
    const int D = ns == 4 ? 512 : 384, FF = 4 * D, S = ns;
    const int ch[4] = {48, 96, 192, 384};
    auto dconv = [&](const std::string &p, int C) {
        for (int j = 0; j < 2; ++j)
        {
            std::string s = p + ".dconv.layers." + std::to_string(j) + ".";
            exp[s + "0.weight"] = (i64)(C / 8) * C * 3;
            exp[s + "0.bias"] = C / 8;
            exp[s + "1.weight"] = C / 8;
            exp[s + "1.bias"] = C / 8;
            exp[s + "3.weight"] = (i64)2 * C * (C / 8);
            exp[s + "3.bias"] = 2 * C;
            exp[s + "4.weight"] = 2 * C;
            exp[s + "4.bias"] = 2 * C;
            exp[s + "6.scale"] = C;
        }
    };
    for (int i = 0; i < 4; ++i)
    {
        int C = ch[i], cf = i == 0 ? 4 : ch[i - 1], ct = i == 0 ? 2 : ch[i - 1];
        std::string e = "encoder." + std::to_string(i), t = "tencoder." + std::to_string(i);
        exp[e + ".conv.weight"] = (i64)C * cf * 8;
        exp[e + ".conv.bias"] = C;
        exp[e + ".rewrite.weight"] = (i64)2 * C * C;
        exp[e + ".rewrite.bias"] = 2 * C;
        dconv(e, C);
        exp[t + ".conv.weight"] = (i64)C * ct * 8;
        exp[t + ".conv.bias"] = C;
        exp[t + ".rewrite.weight"] = (i64)2 * C * C;
        exp[t + ".rewrite.bias"] = 2 * C;
        dconv(t, C);
    }
    for (int k = 0; k < 4; ++k)
    {
        int Cd = ch[3 - k], cf = k < 3 ? ch[2 - k] : 4 * S, ct = k < 3 ? ch[2 - k] : 2 * S;
        std::string d = "decoder." + std::to_string(k), t = "tdecoder." + std::to_string(k);
        exp[d + ".conv_tr.weight"] = (i64)Cd * cf * 8;
        exp[d + ".conv_tr.bias"] = cf;
        exp[d + ".rewrite.weight"] = (i64)2 * Cd * Cd * 9;
        exp[d + ".rewrite.bias"] = 2 * Cd;
        dconv(d, Cd);
        exp[t + ".conv_tr.weight"] = (i64)Cd * ct * 8;
        exp[t + ".conv_tr.bias"] = ct;
        exp[t + ".rewrite.weight"] = (i64)2 * Cd * Cd * 3;
        exp[t + ".rewrite.bias"] = 2 * Cd;
        dconv(t, Cd);
    }
    exp["freq_emb.embedding.weight"] = 512 * 48;
    if (ns == 4)
    {
        const char *nm[4] = {"channel_upsampler", "channel_downsampler", "channel_upsampler_t", "channel_downsampler_t"};
        for (int i = 0; i < 4; ++i)
        {
            exp[std::string(nm[i]) + ".weight"] = 512 * 384;
            exp[std::string(nm[i]) + ".bias"] = (i % 2 == 0) ? 512 : 384;
        }
    }
    for (const char *nm : {"norm_in", "norm_in_t"})
    {
        exp[std::string("crosstransformer.") + nm + ".weight"] = D;
        exp[std::string("crosstransformer.") + nm + ".bias"] = D;
    }
    for (int layer = 0; layer < 5; ++layer)
        for (const char *sfx : {"", "_t"})
        {
            std::string p = std::string("crosstransformer.layers") + sfx + "." + std::to_string(layer);
            std::string a = p + (layer % 2 == 0 ? ".self_attn" : ".cross_attn");
            exp[a + ".in_proj_weight"] = (i64)3 * D * D;
            exp[a + ".in_proj_bias"] = 3 * D;
            exp[a + ".out_proj.weight"] = (i64)D * D;
            exp[a + ".out_proj.bias"] = D;
            exp[p + ".linear1.weight"] = (i64)FF * D;
            exp[p + ".linear1.bias"] = FF;
            exp[p + ".linear2.weight"] = (i64)D * FF;
            exp[p + ".linear2.bias"] = D;
            for (const char *n : {"norm1", "norm2", "norm_out"})
            {
                exp[p + "." + n + ".weight"] = D;
                exp[p + "." + n + ".bias"] = D;
            }
            if (layer % 2 == 1)
            {
                exp[p + ".norm3.weight"] = D;
                exp[p + ".norm3.bias"] = D;
            }
            exp[p + ".gamma_1.scale"] = D;
            exp[p + ".gamma_2.scale"] = D;
        }
    return true;
}

// Demucs v3 hdemucs_mmi: /root/reference/src/model.hpp:694-1236 (shapes), src/model_load.cpp:1388-2136 (names)
static void expected_shapes_v3(std::map<std::string, i64> &exp)
{
    const int ch[4] = {48, 96, 192, 384};
    auto dconv = [&](const std::string &p, int C) {
        const int H = C / 4;
        for (int j = 0; j < 2; ++j)
        {
            std::string s = p + ".dconv.layers." + std::to_string(j) + ".";
            exp[s + "0.weight"] = (i64)H * C * 3;
            exp[s + "0.bias"] = H;
            exp[s + "1.weight"] = H;
            exp[s + "1.bias"] = H;
            exp[s + "3.weight"] = (i64)2 * C * H;
            exp[s + "3.bias"] = 2 * C;
            exp[s + "4.weight"] = 2 * C;
            exp[s + "4.bias"] = 2 * C;
            exp[s + "6.scale"] = C;
        }
    };
    auto dconv_lstm = [&](const std::string &p, int C) {
        const int H = C / 4;
        for (int j = 0; j < 2; ++j)
        {
            std::string s = p + ".dconv.layers." + std::to_string(j) + ".";
            exp[s + "0.weight"] = (i64)H * C * 3;
            exp[s + "0.bias"] = H;
            exp[s + "1.weight"] = H;
            exp[s + "1.bias"] = H;
            for (int layer = 0; layer < 2; ++layer)
                for (const char *sfx : {"", "_reverse"})
                {
                    std::string l = "l" + std::to_string(layer) + sfx;
                    exp[s + "3.lstm.weight_ih_" + l] = (i64)4 * H * (layer == 0 ? H : 2 * H);
                    exp[s + "3.lstm.weight_hh_" + l] = (i64)4 * H * H;
                    exp[s + "3.lstm.bias_ih_" + l] = 4 * H;
                    exp[s + "3.lstm.bias_hh_" + l] = 4 * H;
                }
            exp[s + "3.linear.weight"] = (i64)H * 2 * H;
            exp[s + "3.linear.bias"] = H;
            for (const char *nm : {"content", "query", "key", "proj"})
            {
                exp[s + "4." + nm + ".weight"] = (i64)H * H;
                exp[s + "4." + nm + ".bias"] = H;
            }
            exp[s + "4.query_decay.weight"] = (i64)16 * H;
            exp[s + "4.query_decay.bias"] = 16;
            exp[s + "5.weight"] = (i64)2 * C * H;
            exp[s + "5.bias"] = 2 * C;
            exp[s + "6.weight"] = 2 * C;
            exp[s + "6.bias"] = 2 * C;
            exp[s + "8.scale"] = C;
        }
    };
    for (int i = 0; i < 4; ++i)
    {
        int C = ch[i], cf = i == 0 ? 4 : ch[i - 1], ct = i == 0 ? 2 : ch[i - 1];
        std::string e = "encoder." + std::to_string(i), t = "tencoder." + std::to_string(i);
        exp[e + ".conv.weight"] = (i64)C * cf * 8;
        exp[e + ".conv.bias"] = C;
        exp[e + ".rewrite.weight"] = (i64)2 * C * C;
        exp[e + ".rewrite.bias"] = 2 * C;
        dconv(e, C);
        exp[t + ".conv.weight"] = (i64)C * ct * 8;
        exp[t + ".conv.bias"] = C;
        exp[t + ".rewrite.weight"] = (i64)2 * C * C;
        exp[t + ".rewrite.bias"] = 2 * C;
        dconv(t, C);
    }
    exp["tencoder.4.conv.weight"] = (i64)768 * 384 * 8;
    exp["tencoder.4.conv.bias"] = 768;
    for (int i = 4; i < 6; ++i)
    {
        const int C = i == 4 ? 768 : 1536, cin = C / 2, kt = i == 4 ? 8 : 4;
        std::string e = "encoder." + std::to_string(i);
        exp[e + ".conv.weight"] = (i64)C * cin * kt;
        exp[e + ".conv.bias"] = C;
        exp[e + ".norm1.weight"] = C;
        exp[e + ".norm1.bias"] = C;
        exp[e + ".rewrite.weight"] = (i64)2 * C * C;
        exp[e + ".rewrite.bias"] = 2 * C;
        exp[e + ".norm2.weight"] = 2 * C;
        exp[e + ".norm2.bias"] = 2 * C;
        dconv_lstm(e, C);
    }
    exp["decoder.0.conv_tr.weight"] = (i64)1536 * 768 * 4;
    exp["decoder.0.conv_tr.bias"] = 768;
    exp["decoder.0.norm2.weight"] = 768;
    exp["decoder.0.norm2.bias"] = 768;
    exp["decoder.0.rewrite.weight"] = (i64)3072 * 1536 * 3;
    exp["decoder.0.rewrite.bias"] = 3072;
    exp["decoder.0.norm1.weight"] = 3072;
    exp["decoder.0.norm1.bias"] = 3072;
    exp["decoder.1.conv_tr.weight"] = (i64)768 * 384 * 8;
    exp["decoder.1.conv_tr.bias"] = 384;
    exp["decoder.1.norm2.weight"] = 384;
    exp["decoder.1.norm2.bias"] = 384;
    exp["decoder.1.rewrite.weight"] = (i64)1536 * 768 * 9;
    exp["decoder.1.rewrite.bias"] = 1536;
    exp["decoder.1.norm1.weight"] = 1536;
    exp["decoder.1.norm1.bias"] = 1536;
    exp["tdecoder.0.conv_tr.weight"] = (i64)768 * 384 * 8;
    exp["tdecoder.0.conv_tr.bias"] = 384;
    exp["tdecoder.0.norm2.weight"] = 384;
    exp["tdecoder.0.norm2.bias"] = 384;
    for (int k = 0; k < 4; ++k)
    {
        int Cd = ch[3 - k], cf = k < 3 ? ch[2 - k] : 16, ct = k < 3 ? ch[2 - k] : 8;
        std::string d = "decoder." + std::to_string(k + 2), t = "tdecoder." + std::to_string(k + 1);
        exp[d + ".conv_tr.weight"] = (i64)Cd * cf * 8;
        exp[d + ".conv_tr.bias"] = cf;
        exp[d + ".rewrite.weight"] = (i64)2 * Cd * Cd * 9;
        exp[d + ".rewrite.bias"] = 2 * Cd;
        exp[t + ".conv_tr.weight"] = (i64)Cd * ct * 8;
        exp[t + ".conv_tr.bias"] = ct;
        exp[t + ".rewrite.weight"] = (i64)2 * Cd * Cd * 3;
        exp[t + ".rewrite.bias"] = 2 * Cd;
    }
    exp["freq_emb.embedding.weight"] = 512 * 48;
}

static void pack_v3(PackedModel &pm, const std::map<std::string, Raw> &raw);

bool load_and_pack(const std::string &path, PackedModel &pm, std::string &err)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f)
    {
        err = "load_demucs_model: failed to open " + path;
        return false;
    }
    uint32_t magic = 0;
    if (fread(&magic, 4, 1, f) != 1)
    {
        fclose(f);
        err = "load_demucs_model: invalid model data (short file)";
        return false;
    }
    if (magic == 0x646d6336u)
    {
        pm.n_sources = 6;
        pm.dim = 384;
    }
    else if (magic == 0x646d6334u)
    {
        pm.n_sources = 4;
        pm.dim = 512;
    }
    else if (magic == 0x646d6333u) // "dmc3": Demucs v3 hdemucs_mmi, model_load.cpp:1335-1340
    {
        pm.arch = 3;
        pm.n_sources = 4;
        pm.dim = 0;
    }
    else
    {
        fclose(f);
        err = "load_demucs_model: invalid model data (bad magic)";
        return false;
    }
    std::map<std::string, i64> exp;
    if (pm.arch == 3)
        expected_shapes_v3(exp);
    else
        expected_shapes(pm.n_sources, exp);
    std::map<std::string, Raw> raw;
    for (;;)
    {
        int32_t n_dims = 0, length = 0;
        if (fread(&n_dims, 4, 1, f) != 1)
            break;
        if (fread(&length, 4, 1, f) != 1)
            break;
        if (n_dims < 0 || n_dims > 4 || length <= 0 || length > 1024)
        {
            fclose(f);
            err = "load_demucs_model: corrupt tensor header";
            return false;
        }
        Raw r;
        i64 nel = 1;
        for (int i = 0; i < n_dims; ++i)
        {
            int32_t ne = 0;
            if (fread(&ne, 4, 1, f) != 1)
            {
                fclose(f);
                err = "load_demucs_model: truncated file";
                return false;
            }
            r.shape.push_back(ne);
            nel *= ne;
        }
        std::string name((size_t)length, '\0');
        if (fread(&name[0], 1, (size_t)length, f) != (size_t)length)
        {
            fclose(f);
            err = "load_demucs_model: truncated file";
            return false;
        }
        auto it = exp.find(name);
        if (it == exp.end())
        {
            fclose(f);
            err = "load_demucs_model: failed to load " + name + " (unknown tensor)";
            return false;
        }
        if (it->second != nel)
        {
            fclose(f);
            err = "load_demucs_model: tensor '" + name + "' has wrong size in model file";
            return false;
        }
        std::vector<uint16_t> h((size_t)nel);
        if (fread(h.data(), 2, (size_t)nel, f) != (size_t)nel)
        {
            fclose(f);
            err = "load_demucs_model: truncated tensor data for " + name;
            return false;
        }
        r.d.resize((size_t)nel);
        for (i64 i = 0; i < nel; ++i)
            r.d[(size_t)i] = half_to_float(h[(size_t)i]);
        raw[name] = std::move(r);
        pm.n_tensors++;
    }
    fclose(f);
    for (auto &kv : exp)
        if (!raw.count(kv.first))
        {
            err = "load_demucs_model: tensor '" + kv.first + "' missing from model file";
            return false;
        }

    // ---------------- pack ----------------
    if (pm.arch == 3)
    {
        pack_v3(pm, raw);
        return true;
    }
    Packer P(pm, raw);
    const int ch[4] = {48, 96, 192, 384};
    const int S = pm.n_sources, D = pm.dim;
    std::vector<int> taps8 = {0, 1, 2, 3, 4, 5, 6, 7}, taps3 = {0, 1, 2}, tap1 = {0};
    for (int i = 0; i < 4; ++i)
    {
        int C = ch[i], cf = i == 0 ? 4 : ch[i - 1], ct = i == 0 ? 2 : ch[i - 1];
        for (int br = 0; br < 2; ++br)
        {
            std::string p = std::string(br == 0 ? "encoder." : "tencoder.") + std::to_string(i);
            P.conv(p + ".conv.Wt", p + ".conv.weight", C, br == 0 ? cf : ct, taps8, false);
            P.vec(p + ".conv.b", p + ".conv.bias", rup(C, 16));
            P.dconv(p, C, 8, br == 0);
            P.conv(p + ".rewrite.Wt", p + ".rewrite.weight", 2 * C, C, tap1, true);
            P.vec_paired(p + ".rewrite.b", p + ".rewrite.bias");
        }
    }
    for (int k = 0; k < 4; ++k)
    {
        int Cd = ch[3 - k];
        {
            std::string p = "decoder." + std::to_string(k);
            // 3x3 over (F=kh, T=kw); our K order is (s1 = kw over T, then kh over F, then ci)
            std::vector<int> tm;
            for (int kw = 0; kw < 3; ++kw)
                for (int kh = 0; kh < 3; ++kh)
                    tm.push_back(kh * 3 + kw);
            P.conv(p + ".rewrite.Wt", p + ".rewrite.weight", 2 * Cd, Cd, tm, true);
            P.vec_paired(p + ".rewrite.b", p + ".rewrite.bias");
            P.dconv(p, Cd, 8, true);
            P.conv_tr(p + ".conv_tr.Wt", p + ".conv_tr.b", p + ".conv_tr.weight", p + ".conv_tr.bias");
        }
        {
            std::string p = "tdecoder." + std::to_string(k);
            P.conv(p + ".rewrite.Wt", p + ".rewrite.weight", 2 * Cd, Cd, taps3, true);
            P.vec_paired(p + ".rewrite.b", p + ".rewrite.bias");
            P.dconv(p, Cd);
            P.conv_tr(p + ".conv_tr.Wt", p + ".conv_tr.b", p + ".conv_tr.weight", p + ".conv_tr.bias");
        }
    }
    (void)S;
    P.vec("freq_emb.table", "freq_emb.embedding.weight", 512 * 48);
    if (pm.n_sources == 4)
        for (const char *nm : {"channel_upsampler", "channel_downsampler", "channel_upsampler_t", "channel_downsampler_t"})
        {
            const Raw &b = raw.at(std::string(nm) + ".bias");
            int N = (int)b.numel();
            int Cin = (int)(raw.at(std::string(nm) + ".weight").numel() / N);
            P.conv(std::string(nm) + ".Wt", std::string(nm) + ".weight", N, Cin, tap1, false);
            P.vec(std::string(nm) + ".b", std::string(nm) + ".bias", rup(N, 16));
        }
    for (const char *nm : {"norm_in", "norm_in_t"})
    {
        P.vec(std::string("crosstransformer.") + nm + ".w", std::string("crosstransformer.") + nm + ".weight", D);
        P.vec(std::string("crosstransformer.") + nm + ".b", std::string("crosstransformer.") + nm + ".bias", D);
    }
    for (int layer = 0; layer < 5; ++layer)
        for (const char *sfx : {"", "_t"})
        {
            std::string p = std::string("crosstransformer.layers") + sfx + "." + std::to_string(layer);
            std::string a = p + (layer % 2 == 0 ? ".self_attn" : ".cross_attn");
            P.conv(p + ".in_proj.Wt", a + ".in_proj_weight", 3 * D, D, tap1, false);
            P.vec(p + ".in_proj.b", a + ".in_proj_bias", 3 * D);
            P.conv(p + ".out_proj.Wt", a + ".out_proj.weight", D, D, tap1, false);
            P.vec(p + ".out_proj.b", a + ".out_proj.bias", D);
            P.conv(p + ".linear1.Wt", p + ".linear1.weight", 4 * D, D, tap1, false);
            P.vec(p + ".linear1.b", p + ".linear1.bias", 4 * D);
            P.conv(p + ".linear2.Wt", p + ".linear2.weight", D, 4 * D, tap1, false);
            P.vec(p + ".linear2.b", p + ".linear2.bias", D);
            for (const char *n : {"norm1", "norm2", "norm3", "norm_out"})
            {
                if (std::string(n) == "norm3" && layer % 2 == 0)
                    continue;
                P.vec(p + "." + n + ".w", p + "." + n + ".weight", D);
                P.vec(p + "." + n + ".b", p + "." + n + ".bias", D);
            }
            P.vec(p + ".gamma_1", p + ".gamma_1.scale", D);
            P.vec(p + ".gamma_2", p + ".gamma_2.scale", D);
        }
    return true;
}

// Demucs v3: encoders / tencoders 0-3 and decoders 2-5 / tdecoders 1-4 are packed like their v4 counterparts
// (DConv hidden width C/4); levels 4 / 5, decoder 0 / 1 and tdecoder 0 are new (plan_v3.cpp).
static void pack_v3(PackedModel &pm, const std::map<std::string, Raw> &raw)
{
    Packer P(pm, raw);
    const int ch[4] = {48, 96, 192, 384};
    std::vector<int> taps8 = {0, 1, 2, 3, 4, 5, 6, 7}, taps4 = {0, 1, 2, 3}, taps3 = {0, 1, 2}, tap1 = {0};
    for (int i = 0; i < 4; ++i)
    {
        int C = ch[i], cf = i == 0 ? 4 : ch[i - 1], ct = i == 0 ? 2 : ch[i - 1];
        for (int br = 0; br < 2; ++br)
        {
            std::string p = std::string(br == 0 ? "encoder." : "tencoder.") + std::to_string(i);
            P.conv(p + ".conv.Wt", p + ".conv.weight", C, br == 0 ? cf : ct, taps8, false);
            P.vec(p + ".conv.b", p + ".conv.bias", rup(C, 16));
            P.dconv(p, C, 4, br == 0);
            P.conv(p + ".rewrite.Wt", p + ".rewrite.weight", 2 * C, C, tap1, true);
            P.vec_paired(p + ".rewrite.b", p + ".rewrite.bias");
        }
    }
    P.vec("freq_emb.table", "freq_emb.embedding.weight", 512 * 48);
    P.conv("tencoder.4.conv.Wt", "tencoder.4.conv.weight", 768, 384, taps8, false);
    P.vec("tencoder.4.conv.b", "tencoder.4.conv.bias", 768);
    for (int i = 4; i < 6; ++i)
    {
        const int C = i == 4 ? 768 : 1536;
        std::string p = "encoder." + std::to_string(i);
        // encoder.4: Conv2d (8,1) over the 8 frequency rows of a frame = one run of 8*384 floats, k = f*384 + c;
        // encoder.5: Conv1d k4 s2 p1 over time, k = tap*768 + c
        P.conv(p + ".conv.Wt", p + ".conv.weight", C, C / 2, i == 4 ? taps8 : taps4, false);
        P.vec(p + ".conv.b", p + ".conv.bias", C);
        P.vec(p + ".norm1.w", p + ".norm1.weight", C);
        P.vec(p + ".norm1.b", p + ".norm1.bias", C);
        P.dconv_lstm(p, C);
        P.conv(p + ".rewrite.Wt", p + ".rewrite.weight", 2 * C, C, tap1, false);
        P.vec(p + ".rewrite.b", p + ".rewrite.bias", 2 * C);
        P.vec(p + ".norm2.w", p + ".norm2.weight", 2 * C);
        P.vec(p + ".norm2.b", p + ".norm2.bias", 2 * C);
    }
    // decoder.0 (shared): Conv1d k3 (plain channel order: GroupNorm(4) sits between the conv and the GLU)
    P.conv("decoder.0.rewrite.Wt", "decoder.0.rewrite.weight", 3072, 1536, taps3, false);
    P.vec("decoder.0.rewrite.b", "decoder.0.rewrite.bias", 3072);
    P.vec("decoder.0.norm1.w", "decoder.0.norm1.weight", 3072);
    P.vec("decoder.0.norm1.b", "decoder.0.norm1.bias", 3072);
    P.conv_tr("decoder.0.conv_tr.Wt", "decoder.0.conv_tr.b", "decoder.0.conv_tr.weight", "decoder.0.conv_tr.bias", 4);
    P.vec("decoder.0.norm2.w", "decoder.0.norm2.weight", 768);
    P.vec("decoder.0.norm2.b", "decoder.0.norm2.bias", 768);
    // decoder.1: Conv2d 3x3 on ONE frequency row: the taps kh = 0, 2 only ever meet zero padding, so the packed
    // matrix keeps kh = 1 (source tap index kh*3 + kw = 3 + kw), K = 3 (kw over T) x 768
    P.conv("decoder.1.rewrite.Wt", "decoder.1.rewrite.weight", 1536, 768, {3, 4, 5}, false);
    P.vec("decoder.1.rewrite.b", "decoder.1.rewrite.bias", 1536);
    P.vec("decoder.1.norm1.w", "decoder.1.norm1.weight", 1536);
    P.vec("decoder.1.norm1.b", "decoder.1.norm1.bias", 1536);
    P.conv_tr("decoder.1.conv_tr.Wt", "decoder.1.conv_tr.b", "decoder.1.conv_tr.weight", "decoder.1.conv_tr.bias");
    P.vec("decoder.1.norm2.w", "decoder.1.norm2.weight", 384);
    P.vec("decoder.1.norm2.b", "decoder.1.norm2.bias", 384);
    P.conv_tr("tdecoder.0.conv_tr.Wt", "tdecoder.0.conv_tr.b", "tdecoder.0.conv_tr.weight", "tdecoder.0.conv_tr.bias");
    P.vec("tdecoder.0.norm2.w", "tdecoder.0.norm2.weight", 384);
    P.vec("tdecoder.0.norm2.b", "tdecoder.0.norm2.bias", 384);
    for (int k = 0; k < 4; ++k)
    {
        int Cd = ch[3 - k];
        {
            std::string p = "decoder." + std::to_string(k + 2);
            std::vector<int> tm; // 3x3 over (F = kh, T = kw); our K order: s1 = kw over T, then kh over F, then ci
            for (int kw = 0; kw < 3; ++kw)
                for (int kh = 0; kh < 3; ++kh)
                    tm.push_back(kh * 3 + kw);
            P.conv(p + ".rewrite.Wt", p + ".rewrite.weight", 2 * Cd, Cd, tm, true);
            P.vec_paired(p + ".rewrite.b", p + ".rewrite.bias");
            P.conv_tr(p + ".conv_tr.Wt", p + ".conv_tr.b", p + ".conv_tr.weight", p + ".conv_tr.bias");
        }
        {
            std::string p = "tdecoder." + std::to_string(k + 1);
            P.conv(p + ".rewrite.Wt", p + ".rewrite.weight", 2 * Cd, Cd, taps3, true);
            P.vec_paired(p + ".rewrite.b", p + ".rewrite.bias");
            P.conv_tr(p + ".conv_tr.Wt", p + ".conv_tr.b", p + ".conv_tr.weight", p + ".conv_tr.bias");
        }
    }
}

} // namespace dmx
