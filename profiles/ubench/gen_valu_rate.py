#!/usr/bin/env python
"""Generates valu_rate.hip: issue-rate micro-benchmarks of VALU instruction kinds on gfx950 (8 waves per SIMD, blocks of 64
independent instructions on 8 registers, time relative to v_fma_f32).  Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip"""
ops = {
 # table 1: instruction kinds, VGPR operands
 "fma": "v_fma_f32 {d}, {d}, {d}, {a}", "addf": "v_add_f32 {d}, {d}, {a}", "subf": "v_sub_f32 {d}, {d}, {a}", "mulf": "v_mul_f32 {d}, {d}, {a}",
 "fmac": "v_fmac_f32 {d}, {d}, {a}",
 "maxf": "v_max_f32 {d}, {d}, {a}", "minf": "v_min_f32 {d}, {d}, {a}", "min3": "v_min3_f32 {d}, {d}, {a}, {b}", "med3": "v_med3_f32 {d}, {d}, {a}, {b}",
 "addu": "v_add_u32 {d}, {d}, {a}", "subu": "v_sub_u32 {d}, {d}, {a}", "addco": "v_add_co_u32 {d}, vcc, {d}, {a}", "addc": "v_addc_co_u32 {d}, vcc, {d}, {a}, vcc",
 "add3": "v_add3_u32 {d}, {d}, {a}, {b}", "lshladd": "v_lshl_add_u32 {d}, {d}, 1, {a}", "and": "v_and_b32 {d}, {d}, {a}", "or": "v_or_b32 {d}, {d}, {a}",
 "xor": "v_xor_b32 {d}, {d}, {a}", "not": "v_not_b32 {d}, {a}", "andor": "v_and_or_b32 {d}, {d}, {a}, {b}", "or3": "v_or3_b32 {d}, {d}, {a}, {b}",
 "bfe": "v_bfe_u32 {d}, {d}, 3, 5", "bfi": "v_bfi_b32 {d}, {d}, {a}, {b}", "lshl": "v_lshlrev_b32 {d}, 1, {d}", "lshr": "v_lshrrev_b32 {d}, 1, {d}",
 "ashr": "v_ashrrev_i32 {d}, 1, {d}", "mul24": "v_mul_u32_u24 {d}, {d}, {a}", "mad24": "v_mad_u32_u24 {d}, {d}, {a}, {b}", "mullo": "v_mul_lo_u32 {d}, {d}, {a}",
 "mulhi": "v_mul_hi_u32 {d}, {d}, {a}", "maxu": "v_max_u32 {d}, {d}, {a}", "minu": "v_min_u32 {d}, {d}, {a}",
 "cmpu_s": "v_cmp_lt_u32 s[22:23], {d}, {a}", "cmpu_vcc": "v_cmp_lt_u32 vcc, {d}, {a}", "cmpf_vcc": "v_cmp_lt_f32 vcc, {d}, {a}", "cmpx": "v_cmpx_lt_u32 exec, {d}, {a}",
 "cnd_vcc": "v_cndmask_b32 {d}, {d}, {a}, vcc", "cnd_s": "v_cndmask_b32 {d}, {d}, {a}, s[22:23]",
 "cmp_cnd": "v_cmp_lt_u32 vcc, {d}, {a}\\nv_cndmask_b32 {d}, {d}, {a}, vcc", "cmp_cnd_s": "v_cmp_lt_u32 s[22:23], {d}, {a}\\nv_cndmask_b32 {d}, {d}, {a}, s[22:23]",
 "cnd_vcc_e64": "v_cndmask_b32_e64 {d}, {d}, {a}, vcc", "cnd_vcc_salu": "s_mov_b64 vcc, s[24:25]\\nv_cndmask_b32 {d}, {d}, {a}, vcc",
 "cnd_vcc_salu4": "s_mov_b64 vcc, s[24:25]\\nv_cndmask_b32 {d}, {d}, {a}, vcc\\nv_cndmask_b32 {a}, {a}, {b}, vcc\\nv_cndmask_b32 {b}, {b}, {d}, vcc\\nv_cndmask_b32 {d}, {d}, {b}, vcc",
 "cnd_s_salu": "s_mov_b64 s[22:23], s[24:25]\\nv_cndmask_b32 {d}, {d}, {a}, s[22:23]", "cnd_zero": "v_cndmask_b32 {d}, 0, {a}, s[22:23]",
 "cnd_vcc_cmp4": "v_cmp_lt_u32 vcc, {d}, {a}\\nv_cndmask_b32 {d}, {d}, {a}, vcc\\nv_cndmask_b32 {a}, {a}, {b}, vcc\\nv_cndmask_b32 {b}, {b}, {d}, vcc\\nv_cndmask_b32 {d}, {d}, {b}, vcc",
 "mov": "v_mov_b32 {d}, {a}", "movdpp": "v_mov_b32_dpp {d}, {a} row_shr:1 row_mask:0xf bank_mask:0xf", "adddpp": "v_add_f32_dpp {d}, {a}, {d} row_shr:1 row_mask:0xf bank_mask:0xf",
 "sdwa_cmp": "v_cmp_le_u32_sdwa vcc, {d}, {a} src0_sel:WORD_0 src1_sel:WORD_1", "sdwa_add": "v_add_u32_sdwa {d}, {d}, {a} dst_sel:DWORD src0_sel:WORD_0 src1_sel:WORD_1",
 "pkaddu16": "v_pk_add_u16 {d}, {d}, {a}", "pksubu16c": "v_pk_sub_u16 {d}, {d}, {a} clamp", "pkminu16": "v_pk_min_u16 {d}, {d}, {a}",
 "dot2u16": "v_dot2_u32_u16 {d}, {d}, {a}, {b}", "sad": "v_sad_u32 {d}, {d}, {a}, {b}", "sadu16": "v_sad_u16 {d}, {d}, {a}, {b}",
 "cvtif": "v_cvt_i32_f32 {d}, {a}", "cvtfu": "v_cvt_f32_u32 {d}, {a}", "floor": "v_floor_f32 {d}, {a}", "rndne": "v_rndne_f32 {d}, {a}",
 "rcp": "v_rcp_f32 {d}, {a}", "rsq": "v_rsq_f32 {d}, {a}", "sqrt": "v_sqrt_f32 {d}, {a}", "perm": "v_perm_b32 {d}, {d}, {a}, {b}",
 "readlane": "v_readlane_b32 s20, {a}, 3", "ldexp": "v_ldexp_f32 {d}, {d}, {a}", "mbcnt": "v_mbcnt_lo_u32_b32 {d}, {a}, {d}", "alignbit": "v_alignbit_b32 {d}, {d}, {a}, {b}",
 "bcnt": "v_bcnt_u32_b32 {d}, {a}, {d}", "ffbh": "v_ffbh_u32 {d}, {a}",
 # 16-bit
 "minu16": "v_min_u16 {d}, {d}, {a}", "maxu16": "v_max_u16 {d}, {d}, {a}", "subu16": "v_sub_u16 {d}, {d}, {a}", "addu16": "v_add_u16 {d}, {d}, {a}",
 "mul_lo_u16": "v_mul_lo_u16 {d}, {d}, {a}", "lshl16": "v_lshlrev_b16 {d}, 1, {a}", "lshr16": "v_lshrrev_b16 {d}, 1, {a}", "addf16": "v_add_f16 {d}, {d}, {a}",
 "minf16": "v_min_f16 {d}, {d}, {a}", "mad_u16": "v_mad_u16 {d}, {d}, {a}, {b}",
 # table 2: operand kinds and encodings
 "fma_s": "v_fma_f32 {d}, s20, {d}, {a}", "fma_inl": "v_fma_f32 {d}, 0.5, {d}, {a}", "mul_s": "v_mul_f32 {d}, s20, {a}", "mul_inl": "v_mul_f32 {d}, 0.5, {a}",
 "mul_lit": "v_mul_f32 {d}, 0x3e99999a, {a}", "addf_e64": "v_add_f32_e64 {d}, {d}, {a}", "addf_neg": "v_add_f32_e64 {d}, -{d}, {a}", "addf_s": "v_add_f32 {d}, s20, {a}",
 "addf_inl": "v_add_f32 {d}, 1.0, {a}", "addu_s": "v_add_u32 {d}, s20, {a}", "addu_inl": "v_add_u32 {d}, 4, {a}", "addu_lit": "v_add_u32 {d}, 0x12345, {a}",
 "and_s": "v_and_b32 {d}, s20, {a}", "and_lit": "v_and_b32 {d}, 0xffff, {a}", "lshr_s": "v_lshrrev_b32 {d}, s20, {a}", "lshr_inl": "v_lshrrev_b32 {d}, 3, {a}", "mov_s": "v_mov_b32 {d}, s20",
 "exec_add": "s_mov_b64 exec, s[24:25]\\nv_add_f32 {d}, {d}, {a}", "exec_add_s": "s_mov_b64 exec, s[24:25]\\nv_add_f32 {d}, s20, {a}",
 "salu_mix": "v_add_f32 {d}, {d}, {a}\\ns_and_b64 s[26:27], s[24:25], s[24:25]",
}
pair_ops = {
 # table 3: packed f32 (register pairs): one instruction = two lanes' worth of flops
 "pk_mul_f32": "v_pk_mul_f32 {d}, {d}, {a}", "pk_add_f32": "v_pk_add_f32 {d}, {d}, {a}", "pk_fma_f32": "v_pk_fma_f32 {d}, {d}, {a}, {b}",
 "pk_mul_f32_s": "v_pk_mul_f32 {d}, s[20:21], {a}", "pk_mov": "v_pk_mov_b32 {d}, {a}, {b}", "mov_b64": "v_mov_b64 {d}, {a}",
 "lshl_add_u64": "v_lshl_add_u64 {d}, {a}, 1, {d}",
}
pregs = ["v[10:11]", "v[12:13]", "v[14:15]", "v[16:17]"]
regs = ["v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17"]
clob = '"v10","v11","v12","v13","v14","v15","v16","v17","vcc","s20","s22","s23","s24","s25","s26","s27"'
src = ['// generated by gen_valu_rate.py\n#include <hip/hip_runtime.h>\n#include <cstdio>\n']
for n, t in ops.items():
    body = "\\n".join(t.format(d=regs[i % 8], a=regs[(i + 1) % 8], b=regs[(i + 2) % 8]) for i in range(16)) + "\\n"
    src.append('__global__ void __launch_bounds__(512) k_%s(float* out, int iters)\n{\n    float x = threadIdx.x;\n    const unsigned long long ex = __builtin_amdgcn_read_exec();\n'
               '    asm volatile("v_mov_b32 v10, %%0\\nv_mov_b32 v11, %%0\\nv_mov_b32 v12, %%0\\nv_mov_b32 v13, %%0\\nv_mov_b32 v14, %%0\\nv_mov_b32 v15, %%0\\nv_mov_b32 v16, %%0\\nv_mov_b32 v17, %%0\\n'
               's_mov_b32 s20, 3\\ns_mov_b64 s[24:25], %%1\\ns_mov_b64 s[22:23], %%1" :: "v"(x), "s"(ex) : %s);\n'
               '    for (int i = 0; i < iters; i++)\n    {\n        asm volatile("%s%s%s%s" ::: %s, "exec");\n        asm volatile("s_mov_b64 exec, %%0" :: "s"(ex));\n    }\n'
               '    float r;\n    asm volatile("v_add_f32 %%0, v10, v11" : "=v"(r));\n    out[blockIdx.x * blockDim.x + threadIdx.x] = r;\n}\n' % (n, clob, body, body, body, body, clob))
for n, t in pair_ops.items():
    body = "\\n".join(t.format(d=pregs[i % 4], a=pregs[(i + 1) % 4], b=pregs[(i + 2) % 4]) for i in range(16)) + "\\n"
    src.append('__global__ void __launch_bounds__(512) k_%s(float* out, int iters)\n{\n    float x = threadIdx.x;\n    const unsigned long long ex = __builtin_amdgcn_read_exec();\n'
               '    asm volatile("v_mov_b32 v10, %%0\\nv_mov_b32 v11, %%0\\nv_mov_b32 v12, %%0\\nv_mov_b32 v13, %%0\\nv_mov_b32 v14, %%0\\nv_mov_b32 v15, %%0\\nv_mov_b32 v16, %%0\\nv_mov_b32 v17, %%0\\n'
               's_mov_b32 s20, 3\\ns_mov_b32 s21, 3\\ns_mov_b64 s[24:25], %%1\\ns_mov_b64 s[22:23], %%1" :: "v"(x), "s"(ex) : %s);\n'
               '    for (int i = 0; i < iters; i++)\n    {\n        asm volatile("%s%s%s%s" ::: %s, "exec");\n        asm volatile("s_mov_b64 exec, %%0" :: "s"(ex));\n    }\n'
               '    float r;\n    asm volatile("v_add_f32 %%0, v10, v11" : "=v"(r));\n    out[blockIdx.x * blockDim.x + threadIdx.x] = r;\n}\n' % (n, clob.replace('"s20"','"s20","s21"'), body, body, body, body, clob.replace('"s20"','"s20","s21"')))
src.append('int main()\n{\n    float* out;\n    (void)hipMalloc(&out, 256 * 8 * 512 * 4);\n    hipEvent_t a, b;\n    (void)hipEventCreate(&a);\n    (void)hipEventCreate(&b);\n    const int iters = 2000;\n    float base = 0;\n'
           '#define RUN(NAME) { hipLaunchKernelGGL(k_##NAME, dim3(1024), dim3(512), 0, 0, out, 10); (void)hipEventRecord(a); hipLaunchKernelGGL(k_##NAME, dim3(1024), dim3(512), 0, 0, out, iters); (void)hipEventRecord(b); (void)hipEventSynchronize(b); float ms; (void)hipEventElapsedTime(&ms, a, b); if (base == 0) base = ms; printf("%-12s %.3f ms  %.2f x v_fma_f32\\n", #NAME, ms, ms / base); }\n')
for n in list(ops) + list(pair_ops):
    src.append('    RUN(%s)\n' % n)
src.append('    return 0;\n}\n')
open(__file__.replace("gen_valu_rate.py", "valu_rate.hip"), "w").write("".join(src))
