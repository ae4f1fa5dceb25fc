#!/usr/bin/env python
"""Ablation copies of csrc/igemm_split.hip for the wide direct-fragment kernel igemm_split_linw_kernel (never the product; results wrong).
    planes  : the loop as it would be with PRE-SPLIT activation planes in memory - per block and K-tile three 16-byte loads (the
              third re-reads 16 bytes of the next K-tile: the same request count, a smaller footprint than real 6-byte planes)
              whose raw bits serve as the three planes, no split arithmetic at all
    nosplit : the two loads as they are, the raw bits as planes (split arithmetic removed, loads unchanged)
Writes build/variants/igemm_split_<name>.hip; build with
    make variant1src NAME=linwplanes FILE=igemm_split SRC=build/variants/igemm_split_planes.hip"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = os.path.join(ROOT, "build", "variants")
os.makedirs(out, exist_ok=True)
src = open(os.path.join(ROOT, "demucs_cpp_amd", "csrc", "igemm_split.hip")).read()


def sub(s, old, new):
    assert s.count(old) == 1, old
    return s.replace(old, new)


split = """        constexpr int SET = decltype(setTag)::value;
        const f32x4 lo = aRaw[SET][i][0], hi = aRaw[SET][i][1];
        unsigned h1[4], h2[4], h3[4];
        split3_pk(lo[0], lo[1], h1[0], h2[0], h3[0]);
        split3_pk(lo[2], lo[3], h1[1], h2[1], h3[1]);
        split3_pk(hi[0], hi[1], h1[2], h2[2], h3[2]);
        split3_pk(hi[2], hi[3], h1[3], h2[3], h3[3]);
        u32x4 q1{h1[0], h1[1], h1[2], h1[3]}, q2{h2[0], h2[1], h2[2], h2[3]}, q3{h3[0], h3[1], h3[2], h3[3]};
        asm volatile("" : "+v"(q1), "+v"(q2), "+v"(q3)); // (computed HERE, between the MFMA groups)"""
raw = """        constexpr int SET = decltype(setTag)::value;
        u32x4 q1 = __builtin_bit_cast(u32x4, aRaw[SET][i][0]), q2 = __builtin_bit_cast(u32x4, aRaw[SET][i][1]), q3 = __builtin_bit_cast(u32x4, aRaw3[SET][i]);
        q1 &= 0x3f803f80u, q2 &= 0x3f803f80u, q3 &= 0x3f803f80u; // (finite, small bf16 values whatever the bits were)
        asm volatile("" : "+v"(q1), "+v"(q2), "+v"(q3));"""
s = sub(src, split, raw)
s = sub(s, "    f32x4 aRaw[2][WMF][2];\n    u32x4 aPl[2][WMF][3];\n    auto load_A = [&](auto setTag) {", "    f32x4 aRaw[2][WMF][2], aRaw3[2][WMF];\n    u32x4 aPl[2][WMF][3];\n    auto load_A = [&](auto setTag) {")
lin = """            aRaw[SET][i][0] = *reinterpret_cast<const f32x4 *>(src);
            aRaw[SET][i][1] = *reinterpret_cast<const f32x4 *>(src + 16);
            aOff[i] += KT * 4;"""
gen = """                aRaw[SET][i][0] = *reinterpret_cast<const f32x4 *>(src);
                aRaw[SET][i][1] = *reinterpret_cast<const f32x4 *>(src + 4);"""
head, tail = s.split("template <int WNF, int EPI, bool GEN>\n__global__", 1)  # (the 128-wide linear kernel above holds the same load text)
for name, third in (("planes", True), ("nosplit", False)):
    t = sub(tail, lin, lin.replace("            aOff[i] += KT * 4;", "            aRaw3[SET][i] = %s;\n            aOff[i] += KT * 4;" % ("*reinterpret_cast<const f32x4 *>(src + 128)" if third else "aRaw[SET][i][0]")))
    t = sub(t, gen, gen + "\n                aRaw3[SET][i] = %s;" % ("*reinterpret_cast<const f32x4 *>(src + 32)" if third else "aRaw[SET][i][0]"))
    open(os.path.join(out, "igemm_split_%s.hip" % name), "w").write(head + "template <int WNF, int EPI, bool GEN>\n__global__" + t)
print("wrote igemm_split_planes.hip igemm_split_nosplit.hip")
