// quant_sgemm.c — CPU study of operand-split arithmetics (NOT product code, NOT the oracle): a cblas_sgemm-shaped function
// that quantises the activation operand the way a split GEMM would see it and then multiplies in fp32, bound into the
// oracle through its BLAS hook (oracle_lib.orc_use_blas) by tests/study/split_study.py.
//   QMODE=f32          nothing (the plain fp32 chain of this file: the baseline of the study)
//   QMODE=bf16x3       a = a1 + a2 + a3 (bf16, round to nearest), w = w1 + w2; the dropped a3 w2 is subtracted
//   QMODE=fp16x3row    a 2^s = h1 + h2 + h3 (fp16, round to nearest), s per ROW from the row's max; w one exact fp16 term
//   QMODE=fp16x3mat    the same with one s per call (stands in for "per batch item")
//   QMODE=fp16x3       s = 0
// Calls whose K is a head dimension (QK^T: both operands are activations) are left alone.
#include <immintrin.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static float bf16_rn(float x)
{
    uint32_t u;
    memcpy(&u, &x, 4);
    if ((u & 0x7f800000u) == 0x7f800000u)
        return x;
    u += 0x7fffu + ((u >> 16) & 1u);
    u &= 0xffff0000u;
    float r;
    memcpy(&r, &u, 4);
    return r;
}
static float fp16_rn(float x) { return _cvtsh_ss(_cvtss_sh(x, _MM_FROUND_TO_NEAREST_INT)); } // F16C: round to nearest even, subnormals kept

static long g_inexact_w = 0, g_calls = 0, g_overflow = 0;
void quant_stats(long *out)
{
    out[0] = g_calls, out[1] = g_inexact_w, out[2] = g_overflow;
    g_calls = g_inexact_w = g_overflow = 0;
}

void quant_cblas_sgemm64(int order, int ta, int tb, int64_t M, int64_t N, int64_t K, float alpha, const float *A, int64_t lda, const float *B,
                         int64_t ldb, float beta, float *C, int64_t ldc)
{
    (void)order, (void)ta, (void)tb, (void)alpha, (void)beta; // RowMajor, NoTrans, Trans, 1, 1 (oracle/demucs_oracle.cpp sgemm_nt)
    const char *mode = getenv("QMODE");
    if (!mode)
        mode = "f32";
    const int attn = (K == 64 || K == 48) && (lda == 512 || lda == 384);
    const int bf = !attn && !strcmp(mode, "bf16x3"), hrow = !attn && !strcmp(mode, "fp16x3row"), hmat = !attn && !strcmp(mode, "fp16x3mat"),
              h0 = !attn && !strcmp(mode, "fp16x3");
    __atomic_add_fetch(&g_calls, 1, __ATOMIC_RELAXED);
    float smat = 1.0f;
    if (hmat)
    {
        float mx = 0.f;
        for (int64_t m = 0; m < M; ++m)
            for (int64_t k = 0; k < K; ++k)
                mx = fmaxf(mx, fabsf(A[m * lda + k]));
        smat = mx > 0.f ? exp2f(14.0f - floorf(log2f(mx))) : 1.0f;
    }
    if (hrow || hmat || h0)
    {
        long bad = 0;
        for (int64_t n = 0; n < N; ++n)
            for (int64_t k = 0; k < K; ++k)
                bad += fp16_rn(B[n * ldb + k]) != B[n * ldb + k];
        __atomic_add_fetch(&g_inexact_w, bad, __ATOMIC_RELAXED);
    }
#pragma omp parallel for schedule(static)
    for (int64_t m = 0; m < M; ++m)
    {
        float *q = (float *)malloc((size_t)K * 4), *a3 = (float *)malloc((size_t)K * 4);
        const float *a = A + m * lda;
        float s = 1.0f;
        if (hrow)
        {
            float mx = 0.f;
            for (int64_t k = 0; k < K; ++k)
                mx = fmaxf(mx, fabsf(a[k]));
            s = mx > 0.f ? exp2f(14.0f - floorf(log2f(mx))) : 1.0f;
        }
        if (hmat)
            s = smat;
        for (int64_t k = 0; k < K; ++k)
        {
            float x = a[k];
            a3[k] = 0.f;
            if (bf)
            {
                const float a1 = bf16_rn(x), a2 = bf16_rn(x - a1);
                a3[k] = bf16_rn(x - a1 - a2);
                x = a1 + a2 + a3[k];
            }
            else if (hrow || hmat || h0)
            {
                const float y = x * s;
                const float h1 = fp16_rn(y), h2 = fp16_rn(y - h1), h3 = fp16_rn(y - h1 - h2);
                if (isinf(h1))
                    __atomic_add_fetch(&g_overflow, 1, __ATOMIC_RELAXED);
                x = ((h1 + h2) + h3) / s; // three fp16 terms: an exact fp32 sum (at most 24 significant bits), the scale a power of two
            }
            q[k] = x;
        }
        for (int64_t n = 0; n < N; ++n)
        {
            const float *w = B + n * ldb;
            float acc = C[m * ldc + n];
            if (bf)
                for (int64_t k = 0; k < K; ++k)
                {
                    const float w1 = bf16_rn(w[k]), w2 = w[k] - w1; // (w2 exact for fp16 weights)
                    acc = fmaf(q[k], w[k], acc);
                    acc -= a3[k] * w2; // the dropped sixth product
                }
            else
                for (int64_t k = 0; k < K; ++k)
                    acc = fmaf(q[k], w[k], acc);
            C[m * ldc + n] = acc;
        }
        free(q);
        free(a3);
    }
}
