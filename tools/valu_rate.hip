// Micro-benchmark: issue rate of the vector instructions the ORB kernels lean on, in wave-instructions per clock per CU
// (gfx950).  Build: hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o tools/valu_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define REP8(x) x x x x x x x x
#define BODY(name, ASM)                                                                        \
  __global__ __launch_bounds__(256) void name(unsigned *sink, int iters) {                     \
    unsigned a0 = threadIdx.x, a1 = a0 * 3u + 1u, a2 = a0 * 5u + 2u, a3 = a0 * 7u + 3u;        \
    unsigned b = blockIdx.x + 11u, c = threadIdx.x ^ 0x55u;                                    \
    for (int i = 0; i < iters; i++) {                                                          \
      REP8(asm volatile(ASM : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)       \
    }                                                                                          \
    if ((a0 ^ a1 ^ a2 ^ a3) == 0x12345678u) sink[0] = a0;                                      \
  }
// each ASM block = 4 independent instructions
BODY(k_xor, "v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4")
BODY(k_min_i32, "v_min_i32 %0, %0, %4\n v_min_i32 %1, %1, %4\n v_min_i32 %2, %2, %4\n v_min_i32 %3, %3, %4")
BODY(k_min3_i32, "v_min3_i32 %0, %0, %4, %5\n v_min3_i32 %1, %1, %4, %5\n v_min3_i32 %2, %2, %4, %5\n v_min3_i32 %3, %3, %4, %5")
BODY(k_min_f32, "v_min_f32 %0, %0, %4\n v_min_f32 %1, %1, %4\n v_min_f32 %2, %2, %4\n v_min_f32 %3, %3, %4")
BODY(k_min3_f32, "v_min3_f32 %0, %0, %4, %5\n v_min3_f32 %1, %1, %4, %5\n v_min3_f32 %2, %2, %4, %5\n v_min3_f32 %3, %3, %4, %5")
BODY(k_fma_f32, "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5")
BODY(k_sub_f32, "v_sub_f32 %0, %0, %4\n v_sub_f32 %1, %1, %4\n v_sub_f32 %2, %2, %4\n v_sub_f32 %3, %3, %4")
BODY(k_pk_min_u16, "v_pk_min_u16 %0, %0, %4\n v_pk_min_u16 %1, %1, %4\n v_pk_min_u16 %2, %2, %4\n v_pk_min_u16 %3, %3, %4")
BODY(k_pk_min_i16, "v_pk_min_i16 %0, %0, %4\n v_pk_min_i16 %1, %1, %4\n v_pk_min_i16 %2, %2, %4\n v_pk_min_i16 %3, %3, %4")
BODY(k_pk_min_f16, "v_pk_min_f16 %0, %0, %4\n v_pk_min_f16 %1, %1, %4\n v_pk_min_f16 %2, %2, %4\n v_pk_min_f16 %3, %3, %4")
BODY(k_min3_i16, "v_min3_i16 %0, %0, %4, %5\n v_min3_i16 %1, %1, %4, %5\n v_min3_i16 %2, %2, %4, %5\n v_min3_i16 %3, %3, %4, %5")
BODY(k_pk_add_u16, "v_pk_add_u16 %0, %0, %4\n v_pk_add_u16 %1, %1, %4\n v_pk_add_u16 %2, %2, %4\n v_pk_add_u16 %3, %3, %4")
BODY(k_pk_mad_u16, "v_pk_mad_u16 %0, %0, %4, %5\n v_pk_mad_u16 %1, %1, %4, %5\n v_pk_mad_u16 %2, %2, %4, %5\n v_pk_mad_u16 %3, %3, %4, %5")
BODY(k_perm, "v_perm_b32 %0, %0, %4, %5\n v_perm_b32 %1, %1, %4, %5\n v_perm_b32 %2, %2, %4, %5\n v_perm_b32 %3, %3, %4, %5")
BODY(k_dot2_u16, "v_dot2_u32_u16 %0, %4, %5, %0\n v_dot2_u32_u16 %1, %4, %5, %1\n v_dot2_u32_u16 %2, %4, %5, %2\n v_dot2_u32_u16 %3, %4, %5, %3")
BODY(k_dot4_u8, "v_dot4_u32_u8 %0, %4, %5, %0\n v_dot4_u32_u8 %1, %4, %5, %1\n v_dot4_u32_u8 %2, %4, %5, %2\n v_dot4_u32_u8 %3, %4, %5, %3")
BODY(k_mad_u24, "v_mad_u32_u24 %0, %0, %4, %5\n v_mad_u32_u24 %1, %1, %4, %5\n v_mad_u32_u24 %2, %2, %4, %5\n v_mad_u32_u24 %3, %3, %4, %5")
BODY(k_mul_lo, "v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4")
BODY(k_bcnt, "v_bcnt_u32_b32 %0, %4, %0\n v_bcnt_u32_b32 %1, %4, %1\n v_bcnt_u32_b32 %2, %4, %2\n v_bcnt_u32_b32 %3, %4, %3")
BODY(k_cvt_ub, "v_cvt_f32_ubyte1 %0, %0\n v_cvt_f32_ubyte1 %1, %1\n v_cvt_f32_ubyte1 %2, %2\n v_cvt_f32_ubyte1 %3, %3")
BODY(k_sad_u8, "v_sad_u8 %0, %4, %5, %0\n v_sad_u8 %1, %4, %5, %1\n v_sad_u8 %2, %4, %5, %2\n v_sad_u8 %3, %4, %5, %3")
BODY(k_lshl_or, "v_lshl_or_b32 %0, %0, 3, %4\n v_lshl_or_b32 %1, %1, 3, %4\n v_lshl_or_b32 %2, %2, 3, %4\n v_lshl_or_b32 %3, %3, 3, %4")
BODY(k_cmp, "v_cmp_lt_u32 vcc, %0, %4\n v_cmp_lt_u32 vcc, %1, %4\n v_cmp_lt_u32 vcc, %2, %4\n v_cmp_lt_u32 vcc, %3, %4")

BODY(k_add_u32, "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4")
BODY(k_sub_u32, "v_sub_u32 %0, %0, %4\n v_sub_u32 %1, %1, %4\n v_sub_u32 %2, %2, %4\n v_sub_u32 %3, %3, %4")
BODY(k_and_b32, "v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %4\n v_and_b32 %2, %2, %4\n v_and_b32 %3, %3, %4")
BODY(k_or_b32, "v_or_b32 %0, %0, %4\n v_or_b32 %1, %1, %4\n v_or_b32 %2, %2, %4\n v_or_b32 %3, %3, %4")
BODY(k_lshlrev, "v_lshlrev_b32 %0, 1, %0\n v_lshlrev_b32 %1, 1, %1\n v_lshlrev_b32 %2, 1, %2\n v_lshlrev_b32 %3, 1, %3")
BODY(k_lshrrev, "v_lshrrev_b32 %0, 1, %0\n v_lshrrev_b32 %1, 1, %1\n v_lshrrev_b32 %2, 1, %2\n v_lshrrev_b32 %3, 1, %3")
BODY(k_max_u32, "v_max_u32 %0, %0, %4\n v_max_u32 %1, %1, %4\n v_max_u32 %2, %2, %4\n v_max_u32 %3, %3, %4")
BODY(k_add_f32, "v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4")
BODY(k_mul_f32, "v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4")
BODY(k_max_f32, "v_max_f32 %0, %0, %4\n v_max_f32 %1, %1, %4\n v_max_f32 %2, %2, %4\n v_max_f32 %3, %3, %4")
BODY(k_mul_u24, "v_mul_u32_u24 %0, %0, %4\n v_mul_u32_u24 %1, %1, %4\n v_mul_u32_u24 %2, %2, %4\n v_mul_u32_u24 %3, %3, %4")
BODY(k_mov, "v_mov_b32 %0, %4\n v_mov_b32 %1, %4\n v_mov_b32 %2, %5\n v_mov_b32 %3, %5")
BODY(k_cndmask, "v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc")
BODY(k_add3, "v_add3_u32 %0, %0, %4, %5\n v_add3_u32 %1, %1, %4, %5\n v_add3_u32 %2, %2, %4, %5\n v_add3_u32 %3, %3, %4, %5")
BODY(k_and_or, "v_and_or_b32 %0, %0, %4, %5\n v_and_or_b32 %1, %1, %4, %5\n v_and_or_b32 %2, %2, %4, %5\n v_and_or_b32 %3, %3, %4, %5")
BODY(k_bfe, "v_bfe_u32 %0, %0, 3, 8\n v_bfe_u32 %1, %1, 3, 8\n v_bfe_u32 %2, %2, 3, 8\n v_bfe_u32 %3, %3, 3, 8")
BODY(k_xor_e64, "v_xor_b32_e64 %0, %0, %4\n v_xor_b32_e64 %1, %1, %4\n v_xor_b32_e64 %2, %2, %4\n v_xor_b32_e64 %3, %3, %4")
BODY(k_sub_f32_sdwa, "v_sub_f32_sdwa %0, %0, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_sub_f32_sdwa %1, %1, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_sub_f32_sdwa %2, %2, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_sub_f32_sdwa %3, %3, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1")
BODY(k_min_u32_sdwa, "v_min_u32_sdwa %0, %0, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_min_u32_sdwa %1, %1, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_min_u32_sdwa %2, %2, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_min_u32_sdwa %3, %3, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1")

// scalar ALU: 4 independent s_add_u32 per block
__global__ __launch_bounds__(256) void k_salu(unsigned *sink, int iters) {
  unsigned a0 = blockIdx.x, a1 = blockIdx.x * 3u + 1u, a2 = blockIdx.x * 5u + 2u, a3 = blockIdx.x * 7u + 3u;
  unsigned b = blockIdx.x + 11u;
  for (int i = 0; i < iters; i++) {
    REP8(asm volatile("s_add_u32 %0, %0, %4\n s_add_u32 %1, %1, %4\n s_add_u32 %2, %2, %4\n s_add_u32 %3, %3, %4"
                      : "+s"(a0), "+s"(a1), "+s"(a2), "+s"(a3) : "s"(b) : "scc");)
  }
  if ((a0 ^ a1 ^ a2 ^ a3) == 0x12345678u) sink[0] = a0;
}
// scalar and vector ALU work interleaved in one wave: does the scalar stream cost vector issue slots?
__global__ __launch_bounds__(256) void k_salu_valu(unsigned *sink, int iters) {
  unsigned a0 = blockIdx.x, a1 = blockIdx.x * 3u + 1u;
  unsigned v0 = threadIdx.x, v1 = threadIdx.x * 3u, c = threadIdx.x ^ 0x55u;
  unsigned b = blockIdx.x + 11u;
  for (int i = 0; i < iters; i++) {
    REP8(asm volatile("s_add_u32 %0, %0, %4\n v_min_u32 %2, %2, %5\n s_add_u32 %1, %1, %4\n v_min_u32 %3, %3, %5"
                      : "+s"(a0), "+s"(a1), "+v"(v0), "+v"(v1) : "s"(b), "v"(c) : "scc");)
  }
  if ((a0 ^ a1 ^ v0 ^ v1) == 0x12345678u) sink[0] = a0;
}

typedef void (*kern_t)(unsigned *, int);

static void run(const char *name, kern_t k, unsigned *sink, int cus, double mhz) {
  const int blocks = cus * 8, iters = 4096;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, sink, 64);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, sink, iters);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double winst = (double)blocks * 4.0 * iters * 32.0;  // wave-instructions
  const double per_clk_cu = winst / (ms * 1e-3) / (mhz * 1e6) / cus;
  printf("%-24s %8.3f ms  %6.3f wave-instr/clk/CU (at %.0f MHz)  %5.1f lanes/clk/CU\n", name, ms, per_clk_cu, mhz,
         per_clk_cu * 64);
}

int main() {
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, 0) != hipSuccess) { fprintf(stderr, "no device\n"); return 1; }
  unsigned *sink;
  hipMalloc(&sink, 64);
  const int cus = p.multiProcessorCount;
  const double mhz = p.clockRate / 1000.0;
  printf("%s: %d CUs, %.0f MHz\n", p.name, cus, mhz);
#define R(k) run(#k, k, sink, cus, mhz)
  R(k_xor); R(k_min_i32); R(k_min3_i32); R(k_min_f32); R(k_min3_f32); R(k_fma_f32); R(k_sub_f32);
  R(k_pk_min_u16); R(k_pk_min_i16); R(k_pk_min_f16); R(k_min3_i16); R(k_pk_add_u16); R(k_pk_mad_u16); R(k_perm);
  R(k_dot2_u16); R(k_dot4_u8); R(k_mad_u24); R(k_mul_lo); R(k_bcnt); R(k_cvt_ub); R(k_sad_u8); R(k_lshl_or); R(k_cmp);
  R(k_add_u32); R(k_sub_u32); R(k_and_b32); R(k_or_b32); R(k_lshlrev); R(k_lshrrev); R(k_max_u32); R(k_add_f32);
  R(k_mul_f32); R(k_max_f32); R(k_mul_u24); R(k_mov); R(k_cndmask); R(k_add3); R(k_and_or); R(k_bfe); R(k_xor_e64);
   R(k_sub_f32_sdwa); R(k_min_u32_sdwa);
  R(k_salu); R(k_salu_valu);
  hipFree(sink);
  return 0;
}
