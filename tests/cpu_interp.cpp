// cpu_interp.cpp — TEST-ONLY CPU interpreter of the execution plan (demucs_cpp_amd/csrc/plan.h).
//
// Purpose: validate, in the build container (no GPU), the host-side half of the product
// - weight repacking (model_pack.cpp) and the op list / descriptors (plan.cpp) - against
// the oracle, and serve as the executable specification of every op the HIP kernels
// implement. It is NOT part of the product: nothing under demucs_cpp_amd/ or cli/ links
// it, and the product fails loudly without a GPU.
#include "../demucs_cpp_amd/csrc/model_pack.cpp"
#include "../demucs_cpp_amd/csrc/plan.cpp"

#include <complex>

using namespace dmx;

namespace
{
inline float gelu(float v) { return 0.5f * v * (1.0f + std::erf(v / std::sqrt(2.0f))); }
inline float sigmoidf(float v) { return 1.0f / (1.0f + std::exp(-v)); }

void fft(std::vector<std::complex<double>> &a, int sign)
{
    int n = (int)a.size();
    for (int i = 1, j = 0; i < n; ++i)
    {
        int bit = n >> 1;
        for (; j & bit; bit >>= 1)
            j ^= bit;
        j ^= bit;
        if (i < j)
            std::swap(a[i], a[j]);
    }
    for (int len = 2; len <= n; len <<= 1)
    {
        double ang = sign * 2.0 * M_PI / len;
        std::complex<double> wl(cos(ang), sin(ang));
        for (int i = 0; i < n; i += len)
        {
            std::complex<double> w(1, 0);
            for (int k = 0; k < len / 2; ++k)
            {
                auto u = a[i + k], v = a[i + k + len / 2] * w;
                a[i + k] = u + v;
                a[i + k + len / 2] = u - v;
                w *= wl;
            }
        }
    }
}

struct Interp
{
    PackedModel pm;
    Plan pl;
    std::vector<float> A;
    const float *W;

    void run_igemm(const IGemm &g)
    {
        const i64 M = (i64)g.B * g.P1 * g.P0;
        const int BN = kTileCfgs[g.cfg].BN;
        const bool paired = g.epi == EPI_GLU || g.epi == EPI_GN_GLU_SCALE_RES;
        const i64 rowLen = (i64)g.L0 * g.Cin;
#pragma omp parallel for schedule(static)
        for (i64 m = 0; m < M; ++m)
        {
            int p0 = (int)(m % g.P0), p1 = (int)((m / g.P0) % g.P1), bb = (int)(m / ((i64)g.P0 * g.P1));
            int grp = bb * g.G0 + (g.G0 > 1 ? p0 : 0);
            std::vector<float> arow((size_t)g.K);
            for (int k = 0; k < g.K; ++k)
            {
                int s1 = k / g.seg0, off = k % g.seg0;
                int in1 = p1 * g.stride1 + s1 * g.dil1 - g.pad1;
                i64 e = (i64)(p0 * g.stride0 - g.pad0) * g.Cin + off;
                float a = 0.f;
                if (in1 >= 0 && in1 < g.L1 && e >= 0 && e < rowLen)
                {
                    a = A[(size_t)(g.x + bb * g.xBatchStride + (i64)in1 * rowLen + e)];
                    if (g.pro == PRO_AFFINE)
                    {
                        const float *st = &A[(size_t)(g.proStats + (i64)bb * 4)];
                        a = (a - st[0]) * st[1];
                    }
                    else if (g.pro == PRO_GN_GELU)
                    {
                        const float *st = &A[(size_t)(g.proStats + (i64)grp * 4)];
                        a = gelu((a - st[0]) * st[1] * W[g.proW_w + k] + W[g.proB_w + k]);
                    }
                }
                arow[(size_t)k] = a;
            }
            std::vector<float> v((size_t)g.N);
            for (int n = 0; n < g.N; ++n)
            {
                const float *wr = W + g.w_w + (i64)n * g.Kp;
                float acc = 0.f;
                for (int k = 0; k < g.K; ++k)
                    acc += arow[(size_t)k] * wr[k];
                v[(size_t)n] = acc + W[g.bias_w + n];
            }
            const i64 yrow = g.y + bb * g.yBatchStride + ((i64)p1 * g.P0 + p0) * g.ldy;
            if (g.epi == EPI_LINEAR)
            {
                for (int n = 0; n < g.N; ++n)
                {
                    float t = v[(size_t)n];
                    if (g.act)
                        t = gelu(t);
                    if (g.res >= 0)
                        t += A[(size_t)(g.res + (yrow - g.y) + n)];
                    v[(size_t)n] = t;
                    A[(size_t)(yrow + n)] = t;
                }
            }
            else if (g.epi == EPI_SCALE_RES)
            {
                for (int n = 0; n < g.N; ++n)
                {
                    float t = A[(size_t)(g.res + (yrow - g.y) + n)] + v[(size_t)n] * W[g.scale_w + n];
                    v[(size_t)n] = t;
                    A[(size_t)(yrow + n)] = t;
                }
            }
            else if (paired)
            {
                int C = g.N / 2;
                const float *st = g.epi == EPI_GN_GLU_SCALE_RES ? &A[(size_t)(g.epiStats + (i64)grp * 4)] : nullptr;
                for (int c = 0; c < C; ++c)
                {
                    int na = (c / 16) * 32 + (c % 16), nb = na + 16;
                    float a = v[(size_t)na], bq = v[(size_t)nb];
                    if (st)
                    {
                        a = (a - st[0]) * st[1] * W[g.epiW_w + na] + W[g.epiB_w + na];
                        bq = (bq - st[0]) * st[1] * W[g.epiW_w + nb] + W[g.epiB_w + nb];
                    }
                    float t = a * sigmoidf(bq);
                    if (g.epi == EPI_GN_GLU_SCALE_RES)
                        t = A[(size_t)(g.res + (yrow - g.y) + c)] + W[g.scale_w + c] * t;
                    else if (g.table_w >= 0)
                        t += g.tableScale * W[g.table_w + (i64)p0 * C + c];
                    A[(size_t)(yrow + c)] = t;
                }
            }
            else if (g.epi == EPI_TRCONV)
            {
                for (int n = 0; n < g.N; ++n)
                {
                    int r = n / g.Cout, co = n % g.Cout;
                    int j = g.trS * p0 + r - g.trOff;
                    if (j < 0 || j >= g.Lout)
                        continue;
                    float t = v[(size_t)n];
                    if (g.act)
                        t = gelu(t);
                    i64 o = bb * g.yBatchStride + ((i64)p1 * g.Lout + j) * g.ldy + co;
                    if (g.res >= 0)
                        t += A[(size_t)(g.res + o)];
                    A[(size_t)(g.y + o)] = t;
                }
            }
            if (g.rowstat >= 0)
                for (int nb = 0; nb < g.NB; ++nb)
                {
                    float s = 0.f, ss = 0.f;
                    for (int n = nb * BN; n < std::min(g.N, (nb + 1) * BN); ++n)
                    {
                        if (g.epi == EPI_STATS_FACT) // factorised statistics: roles of the columns, plan.h
                        {
                            if (n < g.Cout)
                                ss += v[(size_t)n] * v[(size_t)n];
                            else if (n == g.Cout)
                                s += v[(size_t)n];
                            else if (n == g.Cout + 1)
                                ss += 2.0f * v[(size_t)n];
                            continue;
                        }
                        s += v[(size_t)n];
                        ss += v[(size_t)n] * v[(size_t)n];
                    }
                    A[(size_t)(g.rowstat + (m * g.NB + nb) * 2)] = s;
                    A[(size_t)(g.rowstat + (m * g.NB + nb) * 2 + 1)] = ss;
                }
        }
    }

    void run_reduce(const StatsReduce &s)
    {
        int G = s.G0 > 1 ? s.G0 : 1;
        for (int b = 0; b < s.B; ++b)
            for (int gi = 0; gi < G; ++gi)
            {
                double sum = 0, sq = 0;
                for (int r = gi; r < s.R; r += G)
                    for (int nb = 0; nb < s.NB; ++nb)
                    {
                        const float *p = &A[(size_t)(s.rowstat + (((i64)b * s.R + r) * s.NB + nb) * 2)];
                        sum += p[0];
                        sq += p[1];
                    }
                double mean = sum / s.count;
                double var = (sq - s.count * mean * mean) / (s.count - 1.0);
                if (var < 0)
                    var = 0;
                float *o = &A[(size_t)(s.out + ((i64)b * G + gi) * 4)];
                o[0] = (float)mean;
                o[2] = (float)std::sqrt(var);
                o[1] = s.mode == MODE_RSTD ? (float)(1.0 / std::sqrt(var + (double)s.eps))
                                           : (float)(1.0 / (std::sqrt(var) + (double)s.eps));
                o[3] = 0.f;
            }
    }

    void run_stft(const Stft &s)
    {
        const float *win = &A[(size_t)s.window];
        for (int b = 0; b < s.B; ++b)
            for (int t = 0; t < s.T; ++t)
            {
                double sum = 0, sq = 0, sumT = 0, sqT = 0;
                for (int ch = 0; ch < 2; ++ch)
                {
                    std::vector<std::complex<double>> buf(4096);
                    for (int i = 0; i < 4096; ++i)
                    {
                        // model-level symmetric padding (Q2), model_inference.cpp:22-46; reference
                        // frame t+2 starts at padded sample t*1024 (see DESIGN.md STFT note)
                        i64 j = (i64)t * 1024 + i - s.pad;
                        if (j < 0)
                            j = -1 - j;
                        if (j >= s.seg)
                            j = 2 * (i64)s.seg - 1 - j;
                        float x = A[(size_t)(s.mix + ((i64)b * s.seg + j) * 2 + ch)];
                        buf[i] = (double)(x * win[i]);
                    }
                    fft(buf, -1);
                    for (int f = 0; f < 2048; ++f)
                    {
                        float re = (float)(buf[f].real() / 64.0), im = (float)(buf[f].imag() / 64.0);
                        float *o = &A[(size_t)(s.x + (((i64)b * s.T + t) * 2048 + f) * 4 + 2 * ch)];
                        o[0] = re, o[1] = im;
                        sum += re + im;
                        sq += (double)re * re + (double)im * im;
                    }
                    for (i64 j = (i64)t * 1024; j < std::min((i64)s.seg, (i64)(t + 1) * 1024); ++j)
                    {
                        float x = A[(size_t)(s.mix + ((i64)b * s.seg + j) * 2 + ch)];
                        sumT += x;
                        sqT += (double)x * x;
                    }
                }
                A[(size_t)(s.rowstat + ((i64)b * s.T + t) * 2)] = (float)sum;
                A[(size_t)(s.rowstat + ((i64)b * s.T + t) * 2 + 1)] = (float)sq;
                A[(size_t)(s.rowstatT + ((i64)b * s.T + t) * 2)] = (float)sumT;
                A[(size_t)(s.rowstatT + ((i64)b * s.T + t) * 2 + 1)] = (float)sqT;
            }
    }

    void run_ln(const LayerNorm &l)
    {
#pragma omp parallel for schedule(static)
        for (int r = 0; r < l.rows; ++r)
        {
            const float *x = &A[(size_t)(l.x + (i64)r * l.D)];
            double s = 0;
            for (int c = 0; c < l.D; ++c)
                s += x[c];
            float mean = (float)(s / l.D);
            double ss = 0;
            for (int c = 0; c < l.D; ++c)
                ss += ((double)x[c] - mean) * ((double)x[c] - mean);
            float rstd = (float)(1.0 / std::sqrt(ss / (l.D - 1) + (double)l.eps));
            std::vector<float> tmp((size_t)l.D);
            for (int c = 0; c < l.D; ++c)
            {
                float v = (x[c] - mean) * rstd * W[l.w_w + c] + W[l.b_w + c];
                if (l.pe >= 0)
                    v += A[(size_t)(l.pe + (i64)(r % l.rowsPerBatch) * l.D + c)];
                tmp[(size_t)c] = v;
            }
            for (int c = 0; c < l.D; ++c)
                A[(size_t)(l.y + (i64)r * l.D + c)] = tmp[(size_t)c];
        }
    }

    void run_gn(const GnApply &g)
    {
        for (int b = 0; b < g.B; ++b)
            for (i64 i = 0; i < (i64)g.rows * g.C; ++i)
            {
                i64 o = (i64)b * g.rows * g.C + i;
                float v = A[(size_t)(g.x + o)];
                if (g.stats >= 0)
                {
                    const float *st = &A[(size_t)(g.stats + (i64)b * 4)];
                    int c = (int)(i % g.C);
                    v = (v - st[0]) * st[1] * W[g.w_w + c] + W[g.b_w + c];
                }
                if (g.res >= 0)
                    v += A[(size_t)(g.res + o)];
                A[(size_t)(g.y + o)] = v;
            }
    }

    void run_attention(const Attention &a)
    {
#pragma omp parallel for collapse(2) schedule(static)
        for (int b = 0; b < a.B; ++b)
            for (int h = 0; h < a.H; ++h)
            {
                std::vector<float> sc((size_t)a.Tk);
                for (int t = 0; t < a.Tq; ++t)
                {
                    const float *q = &A[(size_t)(a.q + b * a.qBatch + (i64)t * a.ldq + h * a.hs)];
                    float mx = -INFINITY;
                    for (int s = 0; s < a.Tk; ++s)
                    {
                        const float *k = &A[(size_t)(a.k + b * a.kBatch + (i64)s * a.ldk + h * a.hs)];
                        float d = 0;
                        for (int j = 0; j < a.hs; ++j)
                            d += q[j] * k[j];
                        sc[(size_t)s] = d * a.scale;
                        mx = std::max(mx, sc[(size_t)s]);
                    }
                    float sum = 0;
                    for (int s = 0; s < a.Tk; ++s)
                    {
                        sc[(size_t)s] = std::exp(sc[(size_t)s] - mx);
                        sum += sc[(size_t)s];
                    }
                    float *o = &A[(size_t)(a.o + b * a.oBatch + (i64)t * a.ldo + h * a.hs)];
                    for (int j = 0; j < a.hs; ++j)
                        o[j] = 0;
                    for (int s = 0; s < a.Tk; ++s)
                    {
                        const float *v = &A[(size_t)(a.v + b * a.vBatch + (i64)s * a.ldv + h * a.hs)];
                        float p = sc[(size_t)s] / sum;
                        for (int j = 0; j < a.hs; ++j)
                            o[j] += p * v[j];
                    }
                }
            }
    }

    void run_istft(const Istft &s)
    {
        const float *win = &A[(size_t)s.window];
        const int CS = 4 * s.S;
        for (int b = 0; b < s.B; ++b)
        {
            const float *st = &A[(size_t)(s.stats + (i64)b * 4)];
            float mean = st[0], stdv = st[2];
            for (int src = 0; src < s.S; ++src)
                for (int ch = 0; ch < 2; ++ch)
                    for (int t = 0; t < s.T; ++t)
                    {
                        std::vector<std::complex<double>> buf(4096, {0, 0});
                        for (int f = 0; f < 2048; ++f)
                        {
                            const float *x = &A[(size_t)(s.x + (((i64)b * s.T + t) * 2048 + f) * CS + src * 4 + 2 * ch)];
                            float re = stdv * x[0] + mean, im = stdv * x[1] + mean; // model_inference.cpp:380-393
                            buf[f] = std::complex<double>((double)(re * 64.0f), (double)(im * 64.0f));
                        }
                        buf[0] = {buf[0].real(), 0};
                        for (int f = 1; f < 2048; ++f)
                            buf[4096 - f] = std::conj(buf[f]);
                        fft(buf, +1);
                        float *o = &A[(size_t)(s.frames + ((((i64)b * s.S + src) * 2 + ch) * s.T + t) * 4096)];
                        for (int i = 0; i < 4096; ++i)
                            o[i] = (float)buf[i].real() * win[i];
                    }
        }
    }

    void run_ola(const Ola &o)
    {
        for (int b = 0; b < o.B; ++b)
        {
            const float *st = &A[(size_t)(o.statsT + (i64)b * 4)];
            for (int src = 0; src < o.S; ++src)
                for (int ch = 0; ch < 2; ++ch)
                    for (int i = 0; i < o.seg; ++i)
                    {
                        // sample n of the (T+4)-frame overlap-add buffer; reference frame f starts at f*1024;
                        // kept waveform drops 2048 (dsp.cpp:113-116) then `pad` (model_inference.cpp:454-455)
                        i64 n = 2048 + o.pad + i;
                        float acc = 0.f;
                        for (i64 f = n / 1024 - 3; f <= n / 1024; ++f)
                        {
                            i64 t = f - 2; // frames 0,1 and T+2,T+3 are zero (model_inference.cpp:439-444)
                            if (t < 0 || t >= o.T)
                                continue;
                            float yv = A[(size_t)(o.frames + ((((i64)b * o.S + src) * 2 + ch) * o.T + t) * 4096 + (n - f * 1024))];
                            acc += yv * 1.0f / 4096.0f / (A[(size_t)(o.wss + n)] + 1e-8f);
                        }
                        float tb = st[2] * A[(size_t)(o.xt + ((i64)b * o.seg + i) * (2 * o.S) + src * 2 + ch)] + st[0];
                        A[(size_t)(o.out + (((i64)b * o.S + src) * 2 + ch) * o.seg + i)] = acc + tb;
                    }
        }
    }

    // order: 0 = plan order. 1 / 2 = the most skewed interleavings two concurrent streams may produce
    // under the derived cross-stream waits (Op::waitOp): always advance stream (order-1) as far as its
    // waits allow before the other stream takes ONE step. If the derived waits were missing a
    // hazard, one of the two skews would reorder the conflicting ops and change the result.
    std::vector<int> schedule(int order) const
    {
        const int n = (int)pl.ops.size();
        std::vector<int> seq;
        if (order == 0)
        {
            for (int i = 0; i < n; ++i)
                seq.push_back(i);
            return seq;
        }
        std::vector<int> q[2];
        for (int i = 0; i < n; ++i)
            q[pl.ops[i].stream ? 1 : 0].push_back(i);
        size_t pos[2] = {0, 0};
        std::vector<char> done(n, 0);
        const int fav = order - 1;
        auto ready = [&](int s) {
            if (pos[s] >= q[s].size())
                return false;
            const int w = pl.ops[q[s][pos[s]]].waitOp;
            return w < 0 || done[w];
        };
        auto step = [&](int s) {
            const int i = q[s][pos[s]++];
            done[i] = 1;
            seq.push_back(i);
        };
        while (pos[0] < q[0].size() || pos[1] < q[1].size())
        {
            if (ready(fav))
                step(fav);
            else if (ready(1 - fav))
                step(1 - fav);
            else
                return std::vector<int>(); // deadlock: cannot happen (waits only point backwards)
        }
        return seq;
    }

    // ---- Demucs v3 ops (plan.h)
    void run_group_stats(const GroupStats &g)
    {
        const int gs = g.C / g.G;
        for (int b = 0; b < g.B; ++b)
            for (int gi = 0; gi < g.G; ++gi)
            {
                const double cnt = (double)g.rows * gs;
                double sum = 0;
                for (int r = 0; r < g.rows; ++r)
                    for (int c = gi * gs; c < (gi + 1) * gs; ++c)
                        sum += A[(size_t)(g.x + ((i64)b * g.rows + r) * g.C + c)];
                const float mean = (float)(sum / cnt);
                double ss = 0;
                for (int r = 0; r < g.rows; ++r)
                    for (int c = gi * gs; c < (gi + 1) * gs; ++c)
                    {
                        const double d = (double)A[(size_t)(g.x + ((i64)b * g.rows + r) * g.C + c)] - (double)mean;
                        ss += d * d;
                    }
                const double var = ss / (cnt - 1.0);
                float *o = &A[(size_t)(g.out + ((i64)b * g.G + gi) * 4)];
                o[0] = mean;
                o[1] = (float)(1.0 / std::sqrt(var + (double)g.eps));
                o[2] = (float)std::sqrt(var);
                o[3] = 0.f;
            }
    }
    void run_gn_act(const GnAct &g)
    {
        const int gs = g.C / g.G, Co = g.mode == 2 ? g.C / 2 : g.C;
        for (int b = 0; b < g.B; ++b)
            for (int r = 0; r < g.rowsOut; ++r)
            {
                const float *x = &A[(size_t)(g.x + ((i64)b * g.rowsIn + r + g.rowOff) * g.C)];
                auto gn = [&](int c) {
                    const float *st = &A[(size_t)(g.stats + ((i64)b * g.G + c / gs) * 4)];
                    return (x[c] - st[0]) * st[1] * W[g.w_w + c] + W[g.b_w + c];
                };
                std::vector<float> tmp((size_t)Co);
                for (int c = 0; c < Co; ++c)
                {
                    float v = gn(c);
                    if (g.mode == 1)
                        v = gelu(v);
                    else if (g.mode == 2)
                        v = v * sigmoidf(gn(c + Co));
                    if (g.scale_w >= 0)
                        v *= W[g.scale_w + c];
                    if (g.res >= 0)
                        v += A[(size_t)(g.res + ((i64)b * g.rowsOut + r) * Co + c)];
                    tmp[(size_t)c] = v;
                }
                for (int c = 0; c < Co; ++c)
                    A[(size_t)(g.y + ((i64)b * g.rowsOut + r) * Co + c)] = tmp[(size_t)c];
            }
    }
    void run_lstm(const Lstm &l)
    {
        const int H = l.H, T = l.T;
#pragma omp parallel for collapse(2) schedule(static)
        for (int b = 0; b < l.B; ++b)
            for (int dir = 0; dir < 2; ++dir)
            {
                std::vector<float> h((size_t)H, 0.f), c((size_t)H, 0.f), hn((size_t)H);
                const float *U = W + l.whh_w + (i64)dir * 4 * H * H;
                for (int step = 0; step < T; ++step)
                {
                    const int t = dir == 0 ? step : T - 1 - step;
                    const float *xp = &A[(size_t)(l.xproj + ((i64)b * T + t) * 8 * H + (i64)dir * 4 * H)];
                    for (int j = 0; j < H; ++j)
                    {
                        float gt[4];
                        for (int q = 0; q < 4; ++q)
                        {
                            const float *u = U + (i64)(4 * j + q) * H;
                            float acc = xp[4 * j + q]; // the MFMA accumulates onto the input projection, k ascending
                            for (int k = 0; k < H; ++k)
                                acc = std::fmaf(u[k], h[(size_t)k], acc);
                            gt[q] = acc;
                        }
                        const float ig = sigmoidf(gt[0]), fg = sigmoidf(gt[1]), gg = std::tanh(gt[2]), og = sigmoidf(gt[3]);
                        const float cn = fg * c[(size_t)j] + ig * gg;
                        c[(size_t)j] = cn;
                        hn[(size_t)j] = og * std::tanh(cn);
                    }
                    h = hn;
                    for (int j = 0; j < H; ++j)
                        A[(size_t)(l.out + ((i64)b * T + t) * 2 * H + (i64)dir * H + j)] = h[(size_t)j];
                }
            }
    }
    void run_local_attn(const LocalAttn &a)
    {
        const int heads = 4, nd = 4, hd = a.H / heads, T = a.T;
        const float sq = std::sqrt((float)hd);
#pragma omp parallel for collapse(2) schedule(static)
        for (int b = 0; b < a.B; ++b)
            for (int h = 0; h < heads; ++h)
            {
                std::vector<float> w((size_t)T);
                for (int s = 0; s < T; ++s)
                {
                    const float *row = &A[(size_t)(a.qkvd + ((i64)b * T + s) * a.ld)];
                    const float *q = row + h * hd;
                    float dq[4];
                    for (int n = 0; n < nd; ++n)
                        dq[n] = 0.5f * sigmoidf(row[3 * a.H + h * nd + n]);
                    float mx = -INFINITY;
                    for (int t = 0; t < T; ++t)
                    {
                        const float *k = &A[(size_t)(a.qkvd + ((i64)b * T + t) * a.ld + a.H + h * hd)];
                        float dot = 0.f;
                        for (int c = 0; c < hd; ++c)
                            dot += q[c] * k[c];
                        float v = dot / sq, decay = 0.f;
                        const float delta = (float)std::abs(t - s);
                        for (int n = 0; n < nd; ++n)
                            decay += (-(float)(n + 1) * delta / 2.0f) * dq[n];
                        w[(size_t)t] = t != s ? v + decay : -100.0f;
                        mx = std::max(mx, w[(size_t)t]);
                    }
                    float sum = 0.f;
                    for (int t = 0; t < T; ++t)
                    {
                        w[(size_t)t] = std::exp(w[(size_t)t] - mx);
                        sum += w[(size_t)t];
                    }
                    for (int t = 0; t < T; ++t)
                        w[(size_t)t] /= sum;
                    float *o = &A[(size_t)(a.out + ((i64)b * T + s) * a.H + h * hd)];
                    for (int c = 0; c < hd; ++c)
                    {
                        float acc = 0.f;
                        for (int t = 0; t < T; ++t)
                            acc += w[(size_t)t] * A[(size_t)(a.qkvd + ((i64)b * T + t) * a.ld + 2 * a.H + h * hd + c)];
                        o[c] = acc;
                    }
                }
            }
    }

    // OP_DCONV_ROW: the unfused definition (plan.h), statistics in double from the formed tensors (no factor)
    void run_dconv_row(const DconvRow &r)
    {
        const int T = r.T, F = r.F, C = r.C, H = r.hid;
#pragma omp parallel for schedule(static)
        for (i64 row = 0; row < (i64)r.B * F; ++row)
        {
            const int b = (int)(row / F), f = (int)(row % F);
            auto X = [&](int t, int c) -> float & { return A[(size_t)(r.x + (((i64)b * T + t) * F + f) * C + c)]; };
            std::vector<float> h((size_t)T * H), y((size_t)T * 2 * C);
            for (int j = 0; j < 2; ++j)
            {
                const int d = j + 1;
                const float *k1 = W + r.k1_w[j], *k1b = W + r.k1_b[j], *k2 = W + r.k2_w[j], *k2b = W + r.k2_b[j];
                double s = 0, q = 0;
                for (int t = 0; t < T; ++t)
                    for (int n = 0; n < H; ++n)
                    {
                        float acc = 0.f;
                        for (int tap = 0; tap < 3; ++tap)
                        {
                            const int ti = t + (tap - 1) * d;
                            if (ti < 0 || ti >= T)
                                continue;
                            for (int c = 0; c < C; ++c)
                                acc += X(ti, c) * k1[(i64)n * 3 * C + tap * C + c];
                        }
                        const float v = acc + k1b[n];
                        h[(size_t)t * H + n] = v;
                        s += v, q += (double)v * v;
                    }
                double cnt = (double)H * T, mean = s / cnt, var = std::max((q - cnt * mean * mean) / (cnt - 1.0), 0.0);
                const float m1 = (float)mean, r1 = (float)(1.0 / std::sqrt(var + (double)r.eps));
                for (int t = 0; t < T; ++t)
                    for (int n = 0; n < H; ++n)
                        h[(size_t)t * H + n] = gelu((h[(size_t)t * H + n] - m1) * r1 * W[r.gn1_w[j] + n] + W[r.gn1_b[j] + n]);
                s = q = 0;
                for (int t = 0; t < T; ++t)
                    for (int n = 0; n < 2 * C; ++n)
                    {
                        float acc = 0.f;
                        for (int k = 0; k < H; ++k)
                            acc += h[(size_t)t * H + k] * k2[(i64)n * 16 + k];
                        const float v = acc + k2b[n];
                        y[(size_t)t * 2 * C + n] = v;
                        s += v, q += (double)v * v;
                    }
                cnt = 2.0 * C * T, mean = s / cnt, var = std::max((q - cnt * mean * mean) / (cnt - 1.0), 0.0);
                const float m2 = (float)mean, r2 = (float)(1.0 / std::sqrt(var + (double)r.eps));
                for (int t = 0; t < T; ++t)
                    for (int c = 0; c < C; ++c)
                    {
                        const int na = (c / 16) * 32 + c % 16, ng = na + 16; // packed GLU pair (model_pack.cpp paired_row)
                        const float a = (y[(size_t)t * 2 * C + na] - m2) * r2 * W[r.gn2_w[j] + na] + W[r.gn2_b[j] + na];
                        const float g = (y[(size_t)t * 2 * C + ng] - m2) * r2 * W[r.gn2_w[j] + ng] + W[r.gn2_b[j] + ng];
                        X(t, c) += W[r.scale_w[j] + c] * (a * sigmoidf(g));
                    }
            }
        }
    }

    void run(int order = 0)
    {
        for (int idx : schedule(order))
        {
            const Op &op = pl.ops[idx];
            switch (op.kind)
            {
            case OP_IGEMM: run_igemm(op.g); break;
            case OP_STATS_REDUCE: run_reduce(op.sr); break;
            case OP_STFT: run_stft(op.stft); break;
            case OP_LAYERNORM: run_ln(op.ln); break;
            case OP_GN_APPLY: run_gn(op.gn); break;
            case OP_ATTENTION: run_attention(op.at); break;
            case OP_ISTFT: run_istft(op.istft); break;
            case OP_OLA: run_ola(op.ola); break;
            case OP_GROUP_STATS: run_group_stats(op.gs); break;
            case OP_GN_ACT: run_gn_act(op.ga); break;
            case OP_LSTM: run_lstm(op.lstm); break;
            case OP_LOCAL_ATTN: run_local_attn(op.la); break;
            case OP_DCONV_ROW: run_dconv_row(op.dr); break;
            default: break;
            }
        }
    }
};
} // namespace

extern "C"
{
    void *interp_create(const char *model_path, int64_t seg, int B)
    {
        auto *it = new Interp();
        std::string err;
        if (!load_and_pack(model_path, it->pm, err))
        {
            fprintf(stderr, "%s\n", err.c_str());
            delete it;
            return nullptr;
        }
        build_plan(it->pm, seg, B, it->pl);
        it->A.assign((size_t)it->pl.arenaFloats, 0.f);
        std::copy(it->pl.constants.begin(), it->pl.constants.end(), it->A.begin());
        it->W = it->pm.blob.data();
        return it;
    }
    // the plan only (no arena): for tools and tests that look at the op list of big batches
    void *interp_create_plan(const char *model_path, int64_t seg, int B)
    {
        auto *it = new Interp();
        std::string err;
        if (!load_and_pack(model_path, it->pm, err))
        {
            fprintf(stderr, "%s\n", err.c_str());
            delete it;
            return nullptr;
        }
        PlanOpts opts;
        if (const char *e = getenv("DMX_INTERP_GEMM")) // tools/plan_dump.py: the tile choices of the split modes (host only)
            opts.gemm = atoi(e), opts.kvPlanes = opts.gemm != GEMM_F32;
        build_plan(it->pm, seg, B, it->pl, opts);
        it->W = nullptr;
        return it;
    }
    void interp_free(void *h) { delete (Interp *)h; }
    int interp_n_ops(void *h) { return (int)((Interp *)h)->pl.ops.size(); }
    double interp_arena_mb(void *h) { return (double)((Interp *)h)->pl.arenaFloats * 4 / 1e6; }
    // mix [B][seg][2] interleaved -> out [B][S][2][seg]
    // number of cross-stream waits in the plan
    int interp_n_waits(void *h)
    {
        int c = 0;
        for (auto &op : ((Interp *)h)->pl.ops)
            c += op.waitOp >= 0;
        return c;
    }
    void interp_run_order(void *h, const float *mix, float *out, int order)
    {
        Interp *it = (Interp *)h;
        const Plan &pl = it->pl;
        std::memcpy(&it->A[(size_t)pl.mixOff], mix, sizeof(float) * (size_t)(pl.B * pl.geo.seg * 2));
        it->run(order);
        std::memcpy(out, &it->A[(size_t)pl.outOff], sizeof(float) * (size_t)((i64)pl.B * pl.S * 2 * pl.geo.seg));
    }
    void interp_run(void *h, const float *mix, float *out)
    {
        Interp *it = (Interp *)h;
        const Plan &pl = it->pl;
        std::memcpy(&it->A[(size_t)pl.mixOff], mix, sizeof(float) * (size_t)(pl.B * pl.geo.seg * 2));
        it->run();
        std::memcpy(out, &it->A[(size_t)pl.outOff], sizeof(float) * (size_t)((i64)pl.B * pl.S * 2 * pl.geo.seg));
    }
    // tap -> returns ndim, fills shape (batch first), copies data if dst != null
    int interp_tap(void *h, const char *name, int64_t *shape, float *dst)
    {
        Interp *it = (Interp *)h;
        for (auto &op : it->pl.ops)
            if (op.kind == OP_TAP && op.name == name)
            {
                int nd = 0;
                i64 per = 1;
                shape[nd++] = it->pl.B;
                for (int j = 0; j < 4 && op.tap.shape[j] > 0; ++j)
                {
                    shape[nd++] = op.tap.shape[j];
                    per *= op.tap.shape[j];
                }
                if (dst)
                    for (int b = 0; b < it->pl.B; ++b)
                        std::memcpy(dst + b * per, &it->A[(size_t)(op.tap.off + b * op.tap.batchStride)], sizeof(float) * (size_t)per);
                return nd;
            }
        return -1;
    }
}

// lists the (tile cfg, prologue, epilogue, act, rowstat) combinations a plan uses (build tooling)
extern "C" int interp_combos(void *h, int *out, int cap)
{
    Interp *it = (Interp *)h;
    int n = 0;
    for (auto &op : it->pl.ops)
        if (op.kind == OP_IGEMM && n + 5 <= cap)
        {
            out[n++] = op.g.cfg;
            out[n++] = op.g.pro;
            out[n++] = op.g.epi;
            out[n++] = op.g.act;
            out[n++] = op.g.rowstat >= 0;
        }
    return n / 5;
}

// one line per igemm op: name, tile cfg, M, N, K, number of tiles (tools: which tile a batch size gets)
extern "C" int interp_plan_dump(void *h, char *buf, int cap)
{
    Interp *it = (Interp *)h;
    std::string all;
    char line[256];
    for (auto &op : it->pl.ops)
        if (op.kind == OP_IGEMM)
        {
            const IGemm &g = op.g;
            const i64 M = (i64)g.B * g.P1 * g.P0;
            const i64 tiles = ((M + kTileCfgs[g.cfg].BM - 1) / kTileCfgs[g.cfg].BM) * g.NB;
            snprintf(line, sizeof(line), "%s %d %lld %d %d %lld %d %d %d\n", op.name.c_str(), g.cfg, (long long)M, g.N, g.K, (long long)tiles,
                     (int)(g.rowstat >= 0), g.hterms, g.epi);
            all += line;
        }
    if ((int)all.size() + 1 > cap)
        return -1;
    memcpy(buf, all.c_str(), all.size() + 1);
    return (int)all.size();
}
