// Issue cost of single VALU instructions on gfx950, measured the way the render kernel runs them: OCC waves per SIMD on every CU, each wave a long
// run of ONE instruction on eight independent register sets (no dependency stalls).  Prints SIMD cycles per wave-level instruction
// (= elapsed * clock / (instructions per wave * waves per SIMD)).  Decides which rewrites of the node / leaf step can pay (profiles/r3_l_*).
//   hipcc -O2 --offload-arch=gfx950 -o /tmp/valu_rate tools/valu_rate.hip && /tmp/valu_rate [waves per SIMD = 6]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#define KERNEL(NAME, ASM8)                                                                                   \
  __global__ __launch_bounds__(256) void k_##NAME(float* out, int iters, float seed) {                       \
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7; \
    float b = seed * 0.5f, c = seed * 0.25f;                                                                 \
    double d0 = seed, d1 = seed + 1, d2 = seed + 2, d3 = seed + 3, e = seed * 0.5;                           \
    for (int i = 0; i < iters; i++) {                                                                        \
      _Pragma("unroll") for (int r = 0; r < 8; r++) { ASM8 }                                                 \
    }                                                                                                        \
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3) == 12345.678f) out[threadIdx.x] = a0;             \
  }
#define F8(INS) asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc", "s4", "s5", "s6", "s7");
// in the strings: %0..%7 = a0..a7, %8 = b, %9 = c
#define I_FMA(x) "v_fma_f32 %" #x ", %" #x ", %8, %9\n"
#define I_MUL(x) "v_mul_f32 %" #x ", %" #x ", %8\n"
#define I_OR(x) "v_or_b32 %" #x ", %" #x ", %8\n"
#define I_ANDOR(x) "v_and_or_b32 %" #x ", %" #x ", %8, %9\n"
#define I_CVTUB(x) "v_cvt_f32_ubyte1 %" #x ", %" #x "\n"
#define I_MIX(x) "v_fma_mix_f32 %" #x ", %" #x ", %8, %9 op_sel_hi:[1,0,0]\n"
#define I_PERM(x) "v_perm_b32 %" #x ", %" #x ", %8, %9\n"
#define I_CND(x) "v_cndmask_b32 %" #x ", %" #x ", %8, vcc\n"
#define I_MAX3(x) "v_max3_f32 %" #x ", %" #x ", %8, %9\n"
#define I_MIN(x) "v_min_f32 %" #x ", %" #x ", %8\n"
#define I_CMP(x) "v_cmp_gt_f32 vcc, %" #x ", %8\n"
#define I_RCP(x) "v_rcp_f32 %" #x ", %" #x "\n"
#define I_SQRT(x) "v_sqrt_f32 %" #x ", %" #x "\n"
#define I_DIVFIX(x) "v_div_fixup_f32 %" #x ", %" #x ", %8, %9\n"
#define I_DIVFMAS(x) "v_div_fmas_f32 %" #x ", %" #x ", %8, %9\n"
#define I_MULLO(x) "v_mul_lo_u32 %" #x ", %" #x ", %8\n"
#define I_MAD24(x) "v_mad_u32_u24 %" #x ", %" #x ", %8, %9\n"
#define I_ADDU(x) "v_add_u32 %" #x ", %" #x ", %8\n"
#define I_XOR(x) "v_xor_b32 %" #x ", %" #x ", %8\n"
#define I_LSHL(x) "v_lshlrev_b32 %" #x ", 3, %" #x "\n"
#define I_BFE(x) "v_bfe_u32 %" #x ", %" #x ", 4, 8\n"
#define I_MINU(x) "v_min_u32 %" #x ", %" #x ", %8\n"
#define I_MOV(x) "v_mov_b32 %" #x ", %8\n"
#define I_NOP(x) "s_nop 0\n"
#define I_NOP1(x) "s_nop 1\n"
#define I_BITOP(x) "v_bitop3_b32 %" #x ", %" #x ", %8, %9 bitop3:0xc8\n"
#define I_LSHLADD(x) "v_lshl_add_u32 %" #x ", %" #x ", 3, %8\n"
#define I_CVTU32(x) "v_cvt_f32_u32 %" #x ", %" #x "\n"
KERNEL(fma_f32, F8(I_FMA))
KERNEL(mul_f32, F8(I_MUL))
KERNEL(or_b32, F8(I_OR))
KERNEL(and_or_b32, F8(I_ANDOR))
KERNEL(cvt_f32_ubyte1, F8(I_CVTUB))
KERNEL(fma_mix_f32, F8(I_MIX))
KERNEL(perm_b32, F8(I_PERM))
KERNEL(cndmask_vcc, F8(I_CND))
KERNEL(max3_f32, F8(I_MAX3))
KERNEL(min_f32, F8(I_MIN))
KERNEL(cmp_gt_f32, F8(I_CMP))
KERNEL(rcp_f32, F8(I_RCP))
KERNEL(sqrt_f32, F8(I_SQRT))
KERNEL(div_fixup_f32, F8(I_DIVFIX))
KERNEL(div_fmas_f32, F8(I_DIVFMAS))
KERNEL(mul_lo_u32, F8(I_MULLO))
KERNEL(mad_u32_u24, F8(I_MAD24))
KERNEL(add_u32, F8(I_ADDU))
KERNEL(xor_b32, F8(I_XOR))
KERNEL(lshlrev_b32, F8(I_LSHL))
KERNEL(bfe_u32, F8(I_BFE))
KERNEL(min_u32, F8(I_MINU))
KERNEL(mov_b32, F8(I_MOV))
KERNEL(s_nop_0, F8(I_NOP))
KERNEL(s_nop_1, F8(I_NOP1))
KERNEL(bitop3_b32, F8(I_BITOP))
KERNEL(lshl_add_u32, F8(I_LSHLADD))
KERNEL(cvt_f32_u32, F8(I_CVTU32))
#define I_AND(x) "v_and_b32 %" #x ", %" #x ", %8\n"
#define I_ADDF(x) "v_add_f32 %" #x ", %" #x ", %8\n"
#define I_SUBF(x) "v_sub_f32 %" #x ", %" #x ", %8\n"
#define I_MAXF(x) "v_max_f32 %" #x ", %" #x ", %8\n"
#define I_MIN3(x) "v_min3_f32 %" #x ", %" #x ", %8, %9\n"
#define I_MED3(x) "v_med3_f32 %" #x ", %" #x ", %8, %9\n"
#define I_FMAC(x) "v_fmac_f32 %" #x ", %8, %9\n"
#define I_CNDS(x) "v_cndmask_b32_e64 %" #x ", %" #x ", %8, s[4:5]\n"
#define I_ORSDWA(x) "v_or_b32_sdwa %" #x ", %8, %" #x " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
#define I_LSHR(x) "v_lshrrev_b32 %" #x ", 8, %" #x "\n"
#define I_ALIGNBIT(x) "v_alignbit_b32 %" #x ", %" #x ", %8, 8\n"
#define I_ADD3(x) "v_add3_u32 %" #x ", %" #x ", %8, %9\n"
#define I_FFBL(x) "v_ffbl_b32 %" #x ", %" #x "\n"
#define I_CMPU(x) "v_cmp_ne_u32 vcc, %" #x ", %8\n"
#define I_CMPS(x) "v_cmp_gt_f32_e64 s[6:7], %" #x ", %8\n"
#define I_MULHI(x) "v_mul_hi_u32 %" #x ", %" #x ", %8\n"
#define I_MUL24(x) "v_mul_u32_u24 %" #x ", %" #x ", %8\n"
#define I_SUBU(x) "v_sub_u32 %" #x ", %" #x ", %8\n"
#define I_ORB3(x) "v_or3_b32 %" #x ", %" #x ", %8, %9\n"
#define I_LSHLOR(x) "v_lshl_or_b32 %" #x ", %" #x ", 3, %8\n"
#define I_CVTI(x) "v_cvt_i32_f32 %" #x ", %" #x "\n"
#define I_FLOOR(x) "v_floor_f32 %" #x ", %" #x "\n"
#define I_CVTPKFP8(x) "v_cvt_f32_fp8 %" #x ", %" #x "\n"
#define I_MADMIX(x) "v_fma_mix_f32 %" #x ", %" #x ", %8, %9\n"
#define I_CMPCND(x) "v_cmp_gt_f32 vcc, %" #x ", %8\ns_nop 1\nv_cndmask_b32 %" #x ", %" #x ", %9, vcc\n"
#define I_DS(x) "ds_read_b32 %" #x ", %8\n"
KERNEL(and_b32, F8(I_AND))
KERNEL(add_f32, F8(I_ADDF))
KERNEL(sub_f32, F8(I_SUBF))
KERNEL(max_f32, F8(I_MAXF))
KERNEL(min3_f32, F8(I_MIN3))
KERNEL(med3_f32, F8(I_MED3))
KERNEL(fmac_f32, F8(I_FMAC))
KERNEL(cndmask_sgpr, F8(I_CNDS))
KERNEL(or_b32_sdwa_byte1, F8(I_ORSDWA))
KERNEL(lshrrev_b32, F8(I_LSHR))
KERNEL(alignbit_b32, F8(I_ALIGNBIT))
KERNEL(add3_u32, F8(I_ADD3))
KERNEL(ffbl_b32, F8(I_FFBL))
KERNEL(cmp_ne_u32, F8(I_CMPU))
KERNEL(cmp_gt_f32_sgpr, F8(I_CMPS))
KERNEL(mul_hi_u32, F8(I_MULHI))
KERNEL(mul_u32_u24, F8(I_MUL24))
KERNEL(sub_u32, F8(I_SUBU))
KERNEL(or3_b32, F8(I_ORB3))
KERNEL(lshl_or_b32, F8(I_LSHLOR))
KERNEL(cvt_i32_f32, F8(I_CVTI))
KERNEL(floor_f32, F8(I_FLOOR))
KERNEL(cvt_f32_fp8, F8(I_CVTPKFP8))
KERNEL(fma_mix_all_f32, F8(I_MADMIX))
KERNEL(cmp_nop_cndmask, F8(I_CMPCND))
// mixes: do the two classes share one issue slot, or run side by side?  (x = 0..7: even registers get the first instruction, odd the second)
#define M8(I0, I1) asm volatile(I0(0) I1(1) I0(2) I1(3) I0(4) I1(5) I0(6) I1(7) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc", "s4", "s5", "s6", "s7");
KERNEL(mix_fma_max3, M8(I_FMA, I_MAX3))
KERNEL(mix_fma_fmamix, M8(I_FMA, I_MIX))
KERNEL(mix_or_cmp, M8(I_OR, I_CMP))
KERNEL(mix_max3_perm, M8(I_MAX3, I_PERM))
KERNEL(mix_fma_rcp, M8(I_FMA, I_RCP))
KERNEL(mix_max3_rcp, M8(I_MAX3, I_RCP))
// f64: four independent accumulators, two instructions each = 8 per block
#define D8(INS) asm volatile(INS(0) INS(1) INS(2) INS(3) INS(0) INS(1) INS(2) INS(3) : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(e));
#define I_FMA64(x) "v_fma_f64 %" #x ", %" #x ", %4, %4\n"
#define I_MUL64(x) "v_mul_f64 %" #x ", %" #x ", %4\n"
#define I_ADD64(x) "v_add_f64 %" #x ", %" #x ", %4\n"
KERNEL(fma_f64, D8(I_FMA64))
KERNEL(mul_f64, D8(I_MUL64))
KERNEL(add_f64, D8(I_ADD64))
// packed f32: two lanes' worth per instruction
#define I_PKFMA(x) "v_pk_fma_f32 %" #x ", %" #x ", %4, %4\n"
KERNEL(pk_fma_f32, D8(I_PKFMA))
// conversions between the two
#define C8(INS) asm volatile(INS(0, 4) INS(1, 5) INS(2, 6) INS(3, 7) INS(0, 4) INS(1, 5) INS(2, 6) INS(3, 7) : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
#define I_CVT64U(x, y) "v_cvt_f64_u32 %" #x ", %" #y "\n"
#define I_CVT3264(x, y) "v_cvt_f32_f64 %" #y ", %" #x "\n"
KERNEL(cvt_f64_u32, C8(I_CVT64U))
KERNEL(cvt_f32_f64, C8(I_CVT3264))

typedef void (*Kern)(float*, int, float);
struct Entry { const char* name; Kern k; };
#define E(NAME) {#NAME, k_##NAME}

int main(int argc, char** argv) {
  const int occ = argc > 1 ? atoi(argv[1]) : 6;
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  int khz = 0; hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
  float* out; hipMalloc(&out, 4096);
  std::vector<Entry> es = {E(fma_f32), E(mul_f32), E(or_b32), E(and_or_b32), E(cvt_f32_ubyte1), E(fma_mix_f32), E(perm_b32), E(cndmask_vcc), E(max3_f32), E(min_f32), E(cmp_gt_f32),
                           E(rcp_f32), E(sqrt_f32), E(div_fixup_f32), E(div_fmas_f32), E(mul_lo_u32), E(mad_u32_u24), E(add_u32), E(xor_b32), E(lshlrev_b32), E(bfe_u32), E(min_u32),
                           E(mov_b32), E(bitop3_b32), E(lshl_add_u32), E(cvt_f32_u32), E(s_nop_0), E(s_nop_1), E(fma_f64), E(mul_f64), E(add_f64), E(pk_fma_f32), E(cvt_f64_u32), E(cvt_f32_f64),
                           E(and_b32), E(add_f32), E(sub_f32), E(max_f32), E(min3_f32), E(med3_f32), E(fmac_f32), E(cndmask_sgpr), E(or_b32_sdwa_byte1), E(lshrrev_b32), E(alignbit_b32),
                           E(add3_u32), E(ffbl_b32), E(cmp_ne_u32), E(cmp_gt_f32_sgpr), E(mul_hi_u32), E(mul_u32_u24), E(sub_u32), E(or3_b32), E(lshl_or_b32), E(cvt_i32_f32), E(floor_f32),
                           E(cvt_f32_fp8), E(fma_mix_all_f32), E(cmp_nop_cndmask),
                           E(mix_fma_max3), E(mix_fma_fmamix), E(mix_or_cmp), E(mix_max3_perm), E(mix_fma_rcp), E(mix_max3_rcp)};
  const int iters = 20000;
  printf("# %s, %d CUs, %d waves per SIMD, nominal clock %d MHz; 64 instructions per loop iteration (+ ~3 of loop overhead), %d iterations\n", prop.name, cus, occ, khz / 1000, iters);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (const Entry& en : es) {
    hipLaunchKernelGGL(en.k, dim3(cus * occ), dim3(256), 0, 0, out, 100, 1.0f);      // warm up
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(en.k, dim3(cus * occ), dim3(256), 0, 0, out, iters, 1.0f);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    const double instr_per_simd = 64.0 * iters * occ;      // each of the occ waves of a SIMD issues 64 * iters of them
    printf("%-16s %8.3f ms   %.2f SIMD cycles per wave-level instruction (at %d MHz)\n", en.name, best, best * 1e-3 * (khz * 1e3) / instr_per_simd, khz / 1000);
  }
  return 0;
}
