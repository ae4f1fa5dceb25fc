// pk_f32_erratum.hip - which instruction goes wrong? (DESIGN.md "packed fp32 next to 16-bit MFMA", found in round 5.)
//
// tools/micro/fft_mfma_repro.hip shows the product's stft_kernel computing wrong frames while ANOTHER kernel's bf16 / f16 MFMAs
// share its CUs, and not at all when the same source is compiled without packed-fp32 (SLP) vectorisation. This file narrows
// it to single instructions: a victim wave runs a chain of ONE packed-fp32 VALU form (inline asm, so the opcode and its
// modifiers are exactly what is written) and checks every result against the two scalar instructions that define it; an
// aggressor kernel on a second stream runs bare MFMAs. Report: mismatches per (victim form, aggressor kind).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -o tools/micro/pk_f32_erratum tools/micro/pk_f32_erratum.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define CK(x)                                                                                  \
    do                                                                                         \
    {                                                                                          \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess)                                                                  \
        {                                                                                      \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(2);                                                                           \
        }                                                                                      \
    } while (0)

struct Rec
{
    unsigned count, iter, lane, form;
    float got[2], want[2], x[2], w[2], c[2];
    unsigned laneHist[64]; // FORM 40: failing results by lane of the wave
    unsigned kind[4];      // FORM 40: the wrong LOW half equals x.lo + 0 (src1 read as zero) | x.lo + c.lo (op_sel ignored) | something else; [3]: wrong HIGH half
};

__device__ __forceinline__ float sfma(float a, float b, float c)
{
    float r;
    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float smul(float a, float b)
{
    float r;
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float sadd(float a, float b)
{
    float r;
    asm volatile("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}


// ---- generic rows: eight independent instances of ONE instruction form back to back (FORM >= 20), or one isolated instance (FORM >= 40)
#define PK2x8(OP, MODS)                                                                                                                          \
    OP " %0, %8, %9 " MODS "\n " OP " %1, %8, %10 " MODS "\n " OP " %2, %9, %10 " MODS "\n " OP " %3, %9, %8 " MODS "\n " OP " %4, %10, %8 " MODS \
       "\n " OP " %5, %10, %9 " MODS "\n " OP " %6, %8, %8 " MODS "\n " OP " %7, %9, %9 " MODS
#define PK3x8(OP, MODS)                                                                                                                                   \
    OP " %0, %8, %9, %10 " MODS "\n " OP " %1, %8, %10, %9 " MODS "\n " OP " %2, %9, %10, %8 " MODS "\n " OP " %3, %9, %8, %10 " MODS "\n " OP " %4, %10, %8, %9 " MODS \
       "\n " OP " %5, %10, %9, %8 " MODS "\n " OP " %6, %8, %8, %9 " MODS "\n " OP " %7, %9, %9, %10 " MODS
struct FormSpec
{
    int kind;      // 0 add, 1 mul, 2 fma
    int sel[3];    // half of source i feeding the LOW lane
    int selh[3];   // half of source i feeding the HIGH lane
    int neg[3];    // source i negated (both lanes)
};
__device__ __forceinline__ v2f form_expect(const FormSpec f, v2f a, v2f b, v2f d)
{
    const float a0 = f.neg[0] ? -a[f.sel[0]] : a[f.sel[0]], a1 = f.neg[0] ? -a[f.selh[0]] : a[f.selh[0]];
    const float b0 = f.neg[1] ? -b[f.sel[1]] : b[f.sel[1]], b1 = f.neg[1] ? -b[f.selh[1]] : b[f.selh[1]];
    const float d0 = f.neg[2] ? -d[f.sel[2]] : d[f.sel[2]], d1 = f.neg[2] ? -d[f.selh[2]] : d[f.selh[2]];
    if (f.kind == 0)
        return v2f{sadd(a0, b0), sadd(a1, b1)};
    if (f.kind == 1)
        return v2f{smul(a0, b0), smul(a1, b1)};
    return v2f{sfma(a0, b0, d0), sfma(a1, b1, d1)};
}
#define ROW8_2(ASM, ...)                                                                                                    \
    {                                                                                                                         \
        v2f t0, t1, t2, t3, t4, t5, t6, t7;                                                                                   \
        asm volatile(ASM : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7) : "v"(x), "v"(w), "v"(c)); \
        const FormSpec fs __VA_ARGS__;                                                                                             \
        const v2f u0 = form_expect(fs, x, w, c), u1 = form_expect(fs, x, c, w), u2 = form_expect(fs, w, c, x), u3 = form_expect(fs, w, x, c),       \
                  u4 = form_expect(fs, c, x, w), u5 = form_expect(fs, c, w, x), u6 = form_expect(fs, x, x, w), u7 = form_expect(fs, w, w, c);       \
        auto S = [](v2f a, v2f b) { return v2f{sadd(a[0], b[0]), sadd(a[1], b[1])}; };                                        \
        auto NE = [](v2f a, v2f b) { return __float_as_uint(a[0]) != __float_as_uint(b[0]) || __float_as_uint(a[1]) != __float_as_uint(b[1]); }; \
        r = S(S(S(t0, t1), S(t2, t3)), S(S(t4, t5), S(t6, t7)));                                                              \
        e = S(S(S(u0, u1), S(u2, u3)), S(S(u4, u5), S(u6, u7)));                                                              \
        if (NE(t0, u0) || NE(t1, u1) || NE(t2, u2) || NE(t3, u3) || NE(t4, u4) || NE(t5, u5) || NE(t6, u6) || NE(t7, u7))     \
            r[0] = __uint_as_float(__float_as_uint(e[0]) ^ 1u);                                                               \
    }

// FORM: 0 v_pk_fma_f32 | 1 v_pk_mul_f32 | 2 v_pk_add_f32 | 3 v_pk_fma_f32 op_sel_hi:[1,0,1] | 4 v_pk_add_f32 neg_lo/hi:[0,1]
//       5 v_pk_mul_f32 with an SGPR-pair operand | 6 v_pk_mov_b32 op_sel:[1,0] | 7 scalar control (v_fma_f32 twice)
//       8 v_pk_mul_f32 op_sel:[1,0] op_sel_hi:[0,0] (the complex-multiply cross term) | 9 = 0 with the operands round-tripped through LDS (ds_write_b64 / ds_read_b64)
template <int FORM>
__global__ __launch_bounds__(256) void victim_kernel(const float *seed, Rec *rec, int iters, float su0, float su1)
{
    __shared__ v2f lds[256];
    const int tid = threadIdx.x, gid = blockIdx.x * 256 + tid;
    v2f x{seed[gid * 6 + 0], seed[gid * 6 + 1]};
    const v2f w{seed[gid * 6 + 2], seed[gid * 6 + 3]}; // |w| < 1
    const v2f c{seed[gid * 6 + 4], seed[gid * 6 + 5]};
    const v2f su{su0, su1};
    unsigned ldsAddr = (unsigned)(size_t)(&lds[tid & 254]); // (FORM 15: two consecutive float2 of this wave's own slots)
    (void)ldsAddr;
    unsigned bad = 0, firstIt = 0;
    v2f fg{0, 0}, fw{0, 0}, fx{0, 0};
    for (int it = 0; it < iters; ++it)
    {
        v2f r, e;
        if constexpr (FORM == 9)
        {
            lds[tid] = x;
            __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0)
            x = lds[tid];
        }
        if constexpr (FORM == 0 || FORM == 9)
        {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(w), "v"(c));
            e = v2f{sfma(x[0], w[0], c[0]), sfma(x[1], w[1], c[1])};
        }
        else if constexpr (FORM == 1)
        {
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(w));
            e = v2f{smul(x[0], w[0]), smul(x[1], w[1])};
        }
        else if constexpr (FORM == 2)
        {
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(c));
            e = v2f{sadd(x[0], c[0]), sadd(x[1], c[1])};
        }
        else if constexpr (FORM == 3)
        {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(x), "v"(w), "v"(c));
            e = v2f{sfma(x[0], w[0], c[0]), sfma(x[1], w[0], c[1])};
        }
        else if constexpr (FORM == 4)
        {
            asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(x), "v"(c));
            e = v2f{sadd(x[0], -c[0]), sadd(x[1], -c[1])};
        }
        else if constexpr (FORM == 5)
        {
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(x), "s"(su));
            e = v2f{smul(x[0], su[0]), smul(x[1], su[1])};
        }
        else if constexpr (FORM == 6)
        {
            asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(r) : "v"(x), "v"(c));
            e = v2f{x[1], c[0]};
        }
        else if constexpr (FORM == 7)
        {
            r = v2f{sfma(x[0], w[0], c[0]), sfma(x[1], w[1], c[1])};
            e = v2f{sfma(x[0], w[0], c[0]), sfma(x[1], w[1], c[1])};
        }
        else if constexpr (FORM == 10)
        {
            // eight DEPENDENT packed fmas back to back (result forwarding between packed instructions)
            v2f t = x;
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n"
                         "v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2"
                         : "+v"(t)
                         : "v"(w), "v"(c));
            r = t;
            e = x;
#pragma unroll
            for (int q = 0; q < 8; ++q)
                e = v2f{sfma(e[0], w[0], c[0]), sfma(e[1], w[1], c[1])};
        }
        else if constexpr (FORM == 11)
        {
            // eight INDEPENDENT packed fmas back to back, summed by scalar adds
            v2f t0, t1, t2, t3, t4, t5, t6, t7;
            const v2f y{x[1], x[0]}, z{c[1], w[0]};
            asm volatile("v_pk_fma_f32 %0, %8, %9, %10\n v_pk_fma_f32 %1, %8, %10, %9\n v_pk_fma_f32 %2, %9, %10, %8\n v_pk_fma_f32 %3, %11, %9, %10\n"
                         "v_pk_fma_f32 %4, %11, %10, %12\n v_pk_fma_f32 %5, %12, %9, %8\n v_pk_fma_f32 %6, %8, %12, %11\n v_pk_fma_f32 %7, %11, %12, %9"
                         : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)
                         : "v"(x), "v"(w), "v"(c), "v"(y), "v"(z));
            auto F = [](v2f a, v2f b, v2f d) { return v2f{sfma(a[0], b[0], d[0]), sfma(a[1], b[1], d[1])}; };
            const v2f u0 = F(x, w, c), u1 = F(x, c, w), u2 = F(w, c, x), u3 = F(y, w, c), u4 = F(y, c, z), u5 = F(z, w, x), u6 = F(x, z, y), u7 = F(y, z, w);
            auto S = [](v2f a, v2f b) { return v2f{sadd(a[0], b[0]), sadd(a[1], b[1])}; };
            r = S(S(S(t0, t1), S(t2, t3)), S(S(t4, t5), S(t6, t7)));
            e = S(S(S(u0, u1), S(u2, u3)), S(S(u4, u5), S(u6, u7)));
        }
        else if constexpr (FORM == 12 || FORM == 15)
        {
            // the pair that fails most often inside the product's stft_kernel (tools/micro/pk_bisect.py, instructions 20 / 21):
            // two packed adds back to back on the same operands, second operand's halves swapped (op_sel), the second one negated
            v2f r0, r1;
            if constexpr (FORM == 15)
            {
                // ... with LDS reads in flight, as in the FFT stages (partial lgkmcnt waits around the butterflies)
                lds[tid] = x;
                v2f l0, l1;
                asm volatile("ds_read_b64 %0, %2\n ds_read_b64 %1, %2 offset:8\n s_waitcnt lgkmcnt(1)\n"
                             "v_pk_add_f32 %3, %0, %5 op_sel:[0,1] op_sel_hi:[1,0]\n"
                             "v_pk_add_f32 %4, %0, %5 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n s_waitcnt lgkmcnt(0)"
                             : "=&v"(l0), "=&v"(l1), "+v"(ldsAddr), "=&v"(r0), "=&v"(r1)
                             : "v"(c)
                             : "memory");
                x = l0;
            }
            else
                asm volatile("v_pk_add_f32 %0, %2, %3 op_sel:[0,1] op_sel_hi:[1,0]\n"
                             "v_pk_add_f32 %1, %2, %3 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]"
                             : "=&v"(r0), "=&v"(r1)
                             : "v"(x), "v"(c));
            const v2f e0{sadd(x[0], c[1]), sadd(x[1], c[0])}, e1{sadd(x[0], -c[1]), sadd(x[1], -c[0])};
            r = v2f{sadd(r0[0], r1[1]), sadd(r0[1], r1[0])};
            e = v2f{sadd(e0[0], e1[1]), sadd(e0[1], e1[0])};
            if (__float_as_uint(r0[0]) != __float_as_uint(e0[0]) || __float_as_uint(r0[1]) != __float_as_uint(e0[1]) ||
                __float_as_uint(r1[0]) != __float_as_uint(e1[0]) || __float_as_uint(r1[1]) != __float_as_uint(e1[1]))
                r[0] = __uint_as_float(__float_as_uint(e[0]) ^ 1u); // any half wrong counts
        }
        else if constexpr (FORM == 13)
        {
            // eight independent packed adds with swapped halves, back to back
            v2f t0, t1, t2, t3, t4, t5, t6, t7;
            asm volatile("v_pk_add_f32 %0, %8, %9 op_sel:[0,1] op_sel_hi:[1,0]\n v_pk_add_f32 %1, %8, %10 op_sel:[0,1] op_sel_hi:[1,0]\n"
                         "v_pk_add_f32 %2, %9, %10 op_sel:[0,1] op_sel_hi:[1,0]\n v_pk_add_f32 %3, %9, %8 op_sel:[0,1] op_sel_hi:[1,0]\n"
                         "v_pk_add_f32 %4, %10, %8 op_sel:[0,1] op_sel_hi:[1,0]\n v_pk_add_f32 %5, %10, %9 op_sel:[0,1] op_sel_hi:[1,0]\n"
                         "v_pk_add_f32 %6, %8, %8 op_sel:[0,1] op_sel_hi:[1,0]\n v_pk_add_f32 %7, %9, %9 op_sel:[0,1] op_sel_hi:[1,0]"
                         : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)
                         : "v"(x), "v"(w), "v"(c));
            auto A = [](v2f a, v2f b) { return v2f{sadd(a[0], b[1]), sadd(a[1], b[0])}; };
            const v2f u0 = A(x, w), u1 = A(x, c), u2 = A(w, c), u3 = A(w, x), u4 = A(c, x), u5 = A(c, w), u6 = A(x, x), u7 = A(w, w);
            auto S = [](v2f a, v2f b) { return v2f{sadd(a[0], b[0]), sadd(a[1], b[1])}; };
            r = S(S(S(t0, t1), S(t2, t3)), S(S(t4, t5), S(t6, t7)));
            e = S(S(S(u0, u1), S(u2, u3)), S(S(u4, u5), S(u6, u7)));
        }
        else if constexpr (FORM == 14)
        {
            // the complex multiply as the compiler packs it: two packed muls and a packed fma with half routing, back to back
            v2f t0, t1, t2;
            asm volatile("v_pk_mul_f32 %0, %3, %4 op_sel:[1,0] op_sel_hi:[0,0]\n v_pk_mul_f32 %1, %3, %5 op_sel_hi:[1,0]\n"
                         "v_pk_fma_f32 %2, %3, %5, %0 op_sel_hi:[1,0,1]"
                         : "=&v"(t0), "=&v"(t1), "=&v"(t2)
                         : "v"(x), "v"(w), "v"(c));
            const v2f u0{smul(x[1], w[0]), smul(x[0], w[0])}, u1{smul(x[0], c[0]), smul(x[1], c[0])};
            const v2f u2{sfma(x[0], c[0], u0[0]), sfma(x[1], c[0], u0[1])};
            r = v2f{sadd(t1[0], t2[1]), sadd(t1[1], t2[0])};
            e = v2f{sadd(u1[0], u2[1]), sadd(u1[1], u2[0])};
        }
        else if constexpr (FORM == 20)
            ROW8_2(PK2x8("v_pk_add_f32", ""), {0,{0,0,0},{1,1,1},{0,0,0}})
        else if constexpr (FORM == 21)
            ROW8_2(PK2x8("v_pk_add_f32", "neg_lo:[0,1] neg_hi:[0,1]"), {0,{0,0,0},{1,1,1},{0,1,0}})
        else if constexpr (FORM == 22)
            ROW8_2(PK2x8("v_pk_add_f32", "op_sel_hi:[1,0]"), {0,{0,0,0},{1,0,1},{0,0,0}})
        else if constexpr (FORM == 23)
            ROW8_2(PK2x8("v_pk_mul_f32", "op_sel_hi:[1,0]"), {1,{0,0,0},{1,0,1},{0,0,0}})
        else if constexpr (FORM == 24)
            ROW8_2(PK3x8("v_pk_fma_f32", "op_sel_hi:[1,1,0]"), {2,{0,0,0},{1,1,0},{0,0,0}})
        else if constexpr (FORM == 25)
            ROW8_2(PK2x8("v_pk_add_f32", "op_sel:[0,1]"), {0,{0,1,0},{1,1,1},{0,0,0}})
        else if constexpr (FORM == 26)
            ROW8_2(PK2x8("v_pk_mul_f32", "op_sel:[0,1] op_sel_hi:[1,0]"), {1,{0,1,0},{1,0,1},{0,0,0}})
        else if constexpr (FORM == 27)
            ROW8_2(PK3x8("v_pk_fma_f32", "op_sel:[0,1,0] op_sel_hi:[1,0,1]"), {2,{0,1,0},{1,0,1},{0,0,0}})
        else if constexpr (FORM == 28)
            ROW8_2(PK2x8("v_pk_add_f32", "op_sel:[1,0] op_sel_hi:[0,1]"), {0,{1,0,0},{0,1,1},{0,0,0}})
        else if constexpr (FORM == 29)
            ROW8_2(PK2x8("v_pk_add_f32", "op_sel:[1,1] op_sel_hi:[0,0]"), {0,{1,1,0},{0,0,1},{0,0,0}})
        else if constexpr (FORM == 30)
            ROW8_2(PK3x8("v_pk_fma_f32", "op_sel_hi:[1,0,1]"), {2,{0,0,0},{1,0,1},{0,0,0}})
        else if constexpr (FORM == 31)
            ROW8_2(PK2x8("v_pk_mul_f32", "op_sel:[1,0] op_sel_hi:[0,0]"), {1,{1,0,0},{0,0,1},{0,0,0}})
        else if constexpr (FORM == 40)
        {
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(x), "v"(c));
            e = v2f{sadd(x[0], c[1]), sadd(x[1], c[0])};
            if (__float_as_uint(r[0]) != __float_as_uint(e[0]) || __float_as_uint(r[1]) != __float_as_uint(e[1]))
            {
                atomicAdd(&rec->laneHist[tid & 63], 1u);
                if (__float_as_uint(r[1]) != __float_as_uint(e[1]))
                    atomicAdd(&rec->kind[3], 1u);
                if (__float_as_uint(r[0]) != __float_as_uint(e[0]))
                    atomicAdd(&rec->kind[__float_as_uint(r[0]) == __float_as_uint(sadd(x[0], 0.0f)) ? 0 : (__float_as_uint(r[0]) == __float_as_uint(sadd(x[0], c[0])) ? 1 : 2)], 1u);
            }
        }
        else if constexpr (FORM == 8)
        {
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,0]" : "=v"(r) : "v"(x), "v"(w));
            e = v2f{smul(x[1], w[0]), smul(x[0], w[0])};
        }
        const bool ne = __float_as_uint(r[0]) != __float_as_uint(e[0]) || __float_as_uint(r[1]) != __float_as_uint(e[1]);
        if (ne && !bad)
            firstIt = it, fg = r, fw = e, fx = x;
        bad += ne ? 1u : 0u;
        // next operand: the EXPECTED value folded back into (-2, 2) (a wrong packed result is counted once, not propagated)
        x = v2f{e[0] - 2.0f * truncf(e[0] * 0.5f), e[1] - 2.0f * truncf(e[1] * 0.5f)};
        if (FORM == 6)
            x = v2f{x[0] + w[0], x[1] + w[1]};
    }
    if (bad)
    {
        const unsigned k = atomicAdd(&rec->count, bad);
        if (k == 0)
        {
            rec->iter = firstIt, rec->lane = (unsigned)gid, rec->form = FORM;
            rec->got[0] = fg[0], rec->got[1] = fg[1], rec->want[0] = fw[0], rec->want[1] = fw[1];
            rec->x[0] = fx[0], rec->x[1] = fx[1], rec->w[0] = w[0], rec->w[1] = w[1], rec->c[0] = c[0], rec->c[1] = c[1];
        }
    }
}

// KIND: 1 v_mfma_f32_16x16x32_bf16 | 2 v_mfma_f32_16x16x4_f32 | 3 v_mfma_f32_32x32x16_bf16 | 4 v_mfma_f32_16x16x32_f16 | 5 VALU only
template <int KIND>
__global__ __launch_bounds__(256) void aggressor_kernel(const unsigned *seed, float *sink, int iters)
{
    const int gid = blockIdx.x * 256 + threadIdx.x;
    unsigned u[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
        u[i] = seed[(gid * 8 + i) & 0xfffff];
    v4f acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
        acc[i] = v4f{0.f, 0.f, 0.f, 0.f};
    unsigned ua[4], ub[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
    {
        ua[i] = KIND == 4 ? ((u[i] & 0x83ff83ffu) | 0x38003800u) : ((u[i] & 0x807f807fu) | 0x3f003f00u);
        ub[i] = KIND == 4 ? ((u[4 + i] & 0x83ff83ffu) | 0x3c003c00u) : ((u[4 + i] & 0x807f807fu) | 0x3f803f80u);
    }
    const float fa = __uint_as_float((u[0] & 0x807fffffu) | 0x3f000000u), fb = __uint_as_float((u[1] & 0x807fffffu) | 0x3f800000u);
    for (int it = 0; it < iters; ++it)
    {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < 8; ++i)
            {
                if constexpr (KIND == 1)
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ua), __builtin_bit_cast(bf16x8, ub), acc[i], 0, 0, 0);
                else if constexpr (KIND == 2)
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, acc[i], 0, 0, 0);
                else if constexpr (KIND == 4)
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ua), __builtin_bit_cast(f16x8, ub), acc[i], 0, 0, 0);
                else
                {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        acc[i][c] = sfma(acc[i][c], fa, fb);
                }
            }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c)
                acc[i][c] = smul(acc[i][c], 0.0625f);
    }
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        t += (acc[i][0] + acc[i][1]) + (acc[i][2] + acc[i][3]);
    if (t == 123.456f)
        sink[0] = t;
}

template <int FORM>
static void run_victim(int grid, const float *seed, Rec *rec, int iters, hipStream_t s)
{
    hipLaunchKernelGGL(victim_kernel<FORM>, dim3(grid), dim3(256), 0, s, seed, rec, iters, 0.7431f, -0.3117f);
}
static void victim(int form, int grid, const float *seed, Rec *rec, int iters, hipStream_t s)
{
    switch (form)
    {
    case 0: run_victim<0>(grid, seed, rec, iters, s); break;
    case 1: run_victim<1>(grid, seed, rec, iters, s); break;
    case 2: run_victim<2>(grid, seed, rec, iters, s); break;
    case 3: run_victim<3>(grid, seed, rec, iters, s); break;
    case 4: run_victim<4>(grid, seed, rec, iters, s); break;
    case 5: run_victim<5>(grid, seed, rec, iters, s); break;
    case 6: run_victim<6>(grid, seed, rec, iters, s); break;
    case 7: run_victim<7>(grid, seed, rec, iters, s); break;
    case 8: run_victim<8>(grid, seed, rec, iters, s); break;
    case 9: run_victim<9>(grid, seed, rec, iters, s); break;
    case 10: run_victim<10>(grid, seed, rec, iters, s); break;
    case 11: run_victim<11>(grid, seed, rec, iters, s); break;
    case 12: run_victim<12>(grid, seed, rec, iters, s); break;
    case 13: run_victim<13>(grid, seed, rec, iters, s); break;
    case 14: run_victim<14>(grid, seed, rec, iters, s); break;
    case 15: run_victim<15>(grid, seed, rec, iters, s); break;
    case 20: run_victim<20>(grid, seed, rec, iters, s); break;
    case 21: run_victim<21>(grid, seed, rec, iters, s); break;
    case 22: run_victim<22>(grid, seed, rec, iters, s); break;
    case 23: run_victim<23>(grid, seed, rec, iters, s); break;
    case 24: run_victim<24>(grid, seed, rec, iters, s); break;
    case 25: run_victim<25>(grid, seed, rec, iters, s); break;
    case 26: run_victim<26>(grid, seed, rec, iters, s); break;
    case 27: run_victim<27>(grid, seed, rec, iters, s); break;
    case 28: run_victim<28>(grid, seed, rec, iters, s); break;
    case 29: run_victim<29>(grid, seed, rec, iters, s); break;
    case 30: run_victim<30>(grid, seed, rec, iters, s); break;
    case 31: run_victim<31>(grid, seed, rec, iters, s); break;
    case 40: run_victim<40>(grid, seed, rec, iters, s); break;
    }
}
static void aggressor(int kind, int grid, const unsigned *seed, float *sink, int iters, hipStream_t s)
{
    switch (kind)
    {
    case 1: hipLaunchKernelGGL(aggressor_kernel<1>, dim3(grid), dim3(256), 0, s, seed, sink, iters); break;
    case 2: hipLaunchKernelGGL(aggressor_kernel<2>, dim3(grid), dim3(256), 0, s, seed, sink, iters / 2); break;
    case 4: hipLaunchKernelGGL(aggressor_kernel<4>, dim3(grid), dim3(256), 0, s, seed, sink, iters); break;
    case 5: hipLaunchKernelGGL(aggressor_kernel<5>, dim3(grid), dim3(256), 0, s, seed, sink, iters / 4); break;
    }
}

int main(int argc, char **argv)
{
    const int rounds = argc > 1 ? atoi(argv[1]) : 8;
    const int grid = 1024; // 4 workgroups per CU of each kernel: both fit beside each other on every CU
    std::mt19937 rng(11);
    std::uniform_real_distribution<float> ud(-0.99f, 0.99f);
    std::vector<float> seed((size_t)grid * 256 * 6);
    for (auto &v : seed)
        v = ud(rng);
    std::vector<unsigned> useed(1 << 20);
    for (auto &v : useed)
        v = rng();
    float *dSeed, *dSink;
    unsigned *dU;
    Rec *dRec;
    CK(hipMalloc(&dSeed, seed.size() * 4));
    CK(hipMalloc(&dU, useed.size() * 4));
    CK(hipMalloc(&dSink, 64));
    CK(hipMalloc(&dRec, sizeof(Rec)));
    CK(hipMemcpy(dSeed, seed.data(), seed.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dU, useed.data(), useed.size() * 4, hipMemcpyHostToDevice));
    hipStream_t sv, sa;
    CK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    const char *fname[16] = {"v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_pk_fma_f32 op_sel_hi:[1,0,1]", "v_pk_add_f32 neg:[0,1]", "v_pk_mul_f32 v, s[pair]",
                             "v_pk_mov_b32 op_sel:[1,0]", "scalar v_fma_f32 (control)", "v_pk_mul_f32 op_sel:[1,0] op_sel_hi:[0,0]", "v_pk_fma_f32 after ds_write/read_b64", "8 dependent v_pk_fma_f32 back to back",
                             "8 independent v_pk_fma_f32 back to back", "pair of v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0] (+neg)",
                             "8 v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0]", "packed complex multiply (mul, mul, fma with op_sel)", "the pair, LDS reads in flight"};
    const char *aname[6] = {"none", "mfma_16x16x32_bf16", "mfma_16x16x4_f32", "", "mfma_16x16x32_f16", "valu_fma"};
    const int vit = 4000, ait = 6000;
    printf("%-58s", "victim form \\ aggressor");
    for (int k : {0, 1, 2, 4, 5})
        printf(" %20s", aname[k]);
    printf("   (mismatching results of %ld per cell)\n", (long)rounds * grid * 256 * vit);
    std::vector<std::pair<int, std::string>> forms;
    for (int f = 0; f < 16; ++f)
        forms.push_back({f, fname[f]});
    forms.push_back({20, "v_pk_add_f32 x8 plain (control)"});
    forms.push_back({21, "v_pk_add_f32 x8 neg_lo/hi:[0,1] (control)"});
    forms.push_back({22, "v_pk_add_f32 x8 op_sel_hi:[1,0] (broadcast lo)"});
    forms.push_back({23, "v_pk_mul_f32 x8 op_sel_hi:[1,0] (broadcast lo)"});
    forms.push_back({24, "v_pk_fma_f32 x8 op_sel_hi:[1,1,0] (broadcast lo)"});
    forms.push_back({25, "v_pk_add_f32 x8 op_sel:[0,1] (broadcast hi)"});
    forms.push_back({26, "v_pk_mul_f32 x8 op_sel:[0,1] op_sel_hi:[1,0] (swap)"});
    forms.push_back({27, "v_pk_fma_f32 x8 op_sel:[0,1,0] op_sel_hi:[1,0,1] (swap)"});
    forms.push_back({28, "v_pk_add_f32 x8 op_sel:[1,0] op_sel_hi:[0,1] (swap src0)"});
    forms.push_back({29, "v_pk_add_f32 x8 op_sel:[1,1] op_sel_hi:[0,0] (swap both)"});
    forms.push_back({30, "v_pk_fma_f32 x8 op_sel_hi:[1,0,1]"});
    forms.push_back({31, "v_pk_mul_f32 x8 op_sel:[1,0] op_sel_hi:[0,0]"});
    forms.push_back({40, "v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0], one isolated instance"});
    for (auto &fe : forms)
    {
        const int form = fe.first;
        if (form < (argc > 2 ? atoi(argv[2]) : 0))
            continue;
        printf("%-58s", fe.second.c_str());
        Rec firstRec{};
        bool haveRec = false;
        unsigned long long laneHist[64] = {0}, kindSum[4] = {0};
        for (int kind : {0, 1, 2, 4, 5})
        {
            unsigned long long total = 0;
            for (int r = 0; r < rounds; ++r)
            {
                CK(hipMemset(dRec, 0, sizeof(Rec)));
                if (kind)
                    aggressor(kind, grid, dU, dSink, ait, sa);
                victim(form, grid, dSeed, dRec, vit, sv);
                CK(hipDeviceSynchronize());
                Rec h;
                CK(hipMemcpy(&h, dRec, sizeof h, hipMemcpyDeviceToHost));
                total += h.count;
                if (form == 40 && h.count)
                {
                    for (int l = 0; l < 64; ++l)
                        laneHist[l] += h.laneHist[l];
                    for (int q = 0; q < 4; ++q)
                        kindSum[q] += h.kind[q];
                }
                if (h.count && !haveRec)
                    firstRec = h, haveRec = true;
            }
            printf(" %20llu", total);
            fflush(stdout);
        }
        printf("\n");
        if (form == 40 && haveRec)
        {
            printf("      failing results by 16-lane group of the wave:");
            for (int g = 0; g < 4; ++g)
            {
                unsigned long long t = 0;
                for (int l = 16 * g; l < 16 * g + 16; ++l)
                    t += laneHist[l];
                printf(" lanes %d-%d: %llu", 16 * g, 16 * g + 15, t);
            }
            printf("\n      wrong low half = x.lo + 0 (src1.hi read as zero): %llu, = x.lo + c.lo (op_sel ignored): %llu, other: %llu; wrong high half: %llu\n", kindSum[0],
                   kindSum[1], kindSum[2], kindSum[3]);
        }
        if (haveRec)
            printf("      first: iteration %u, global lane %u (lane %u of its wave): got (%.9g, %.9g) want (%.9g, %.9g); x (%.9g, %.9g) w (%.9g, %.9g) c (%.9g, %.9g)\n",
                   firstRec.iter, firstRec.lane, firstRec.lane & 63, firstRec.got[0], firstRec.got[1], firstRec.want[0], firstRec.want[1], firstRec.x[0],
                   firstRec.x[1], firstRec.w[0], firstRec.w[1], firstRec.c[0], firstRec.c[1]);
    }
    return 0;
}
