// ubench_opcodes.hip -- issue cost per opcode on gfx950 for every instruction of the sketch
// kernel's hot block.  One kernel per opcode: 8 independent register
// chains per lane, 1024-thread workgroups, 4 waves per SIMD (the sketch kernel's occupancy),
// all 256 CUs busy, so the figure is the THROUGHPUT cost in SIMD cycles per wave-instruction
// (kernel time by HIP events) at the clock the chip holds under that load (measured per kernel:
// s_memtime ticks per s_memrealtime tick, the latter at 100 MHz).  The second figure is what
// the oldest wave of a SIMD sees by its own cycle counter: the issue arbiter serves the oldest
// wave first, so it runs at its single-wave speed and finishes early -- not a throughput.
//
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_opcodes.hip -o /tmp/ubench_opcodes && /tmp/ubench_opcodes
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

enum Op {
  ADD_U32, XOR_B32, AND_B32, OR_B32, MOV_B32, LSHLREV_B32, LSHRREV_B32, LSHL_ADD_U32, LSHL_OR_B32, ADD3_U32, AND_OR_B32,
  XAD_U32, BFE_U32, BFI_B32, ALIGNBIT_B32, ALIGNBYTE_B32, PERM_B32, CNDMASK_VCC, CNDMASK_SGPR, CMP_LT_U32, CMP_LT_U64, CMP_LT_U32_SGPR,
  LSHLREV_B64, LSHRREV_B64, MUL_LO_U32, MUL_HI_U32, MAD_U64_U32, MUL_U32_U24, MUL_HI_U32_U24, MAD_U32_U24,
  MBCNT_LO, MBCNT_HI, FFBH_U32, MIN3_U32, MIN_U32, ADD_U32_SDWA, ADD_CO_U32, ADDC_CO_U32, SUB_CLAMP, READFIRSTLANE, READLANE,
  MAD_U32_U16, PK_MUL_LO_U16, PK_ADD_U16, ADD_LSHL_U32, MOV_DPP, DS_WRITE_B64, DS_READ_U8, DS_MIN_U32_RANDOM, DS_READ_B64,
  MUL_LO_U32_LIT, SAD_U32, MAD_U64_U32_ACC, NOP, N_OPS
};

struct OpInfo { const char *name; int wave_insts_per_unit; };

template <int OP>
__global__ __launch_bounds__(1024) void k(uint32_t *out, uint32_t iters, uint32_t seed, uint64_t *clk) {
  __shared__ uint64_t lds64[2048 + 64];
  uint32_t t = threadIdx.x + blockIdx.x * 1024u + seed;
  uint32_t a0 = t, a1 = t * 3 + 1, a2 = t ^ 0x1234567u, a3 = t + 77, a4 = t * 5 + 3, a5 = ~t, a6 = t + 9, a7 = t * 7 + 5;
  uint32_t b = t * 2654435761u | 1u, c = (t >> 3) | 0x10001u;
  uint64_t w0 = t * 0x9E3779B97F4A7C15ULL, w1 = ~w0, w2 = w0 * 3, w3 = w0 + 12345, w4 = w0 ^ 0x5555, w5 = w1 * 7, w6 = w1 + 99, w7 = w0 * 11;
  uint32_t s0 = seed * 13u + 0x6659FD93u;  // wave-uniform: lives in an SGPR
  s0 = __builtin_amdgcn_readfirstlane(s0);
  uint32_t ldsaddr = (threadIdx.x & 1023u) * 8u;  // conflict-free 8-byte stride
  lds64[threadIdx.x] = w0; lds64[threadIdx.x + 1024] = w1;
  __syncthreads();
  uint64_t t0 = 0, r0 = 0;
  if (threadIdx.x == 0) { t0 = __builtin_readcyclecounter(); r0 = wall_clock64(); }
  for (uint32_t i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#define A(n) a##n
#define W(n) w##n
      if (OP == ADD_U32) {
#define X(n) asm volatile("v_add_u32 %0, %0, %1" : "+v"(A(n)) : "v"(b));
        REP8(X)
#undef X
      } else if (OP == XOR_B32) {
#define X(n) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(A(n)) : "v"(b));
        REP8(X)
#undef X
      } else if (OP == AND_B32) {
#define X(n) asm volatile("v_and_b32 %0, %0, %1" : "+v"(A(n)) : "v"(b));
        REP8(X)
#undef X
      } else if (OP == OR_B32) {
#define X(n) asm volatile("v_or_b32 %0, %0, %1" : "+v"(A(n)) : "v"(b));
        REP8(X)
#undef X
      } else if (OP == MOV_B32) {
#define X(n) asm volatile("v_mov_b32 %0, %1" : "+v"(A(n)) : "v"(b));
        REP8(X)
#undef X
      } else if (OP == LSHLREV_B32) {
#define X(n) asm volatile("v_lshlrev_b32 %0, 2, %0" : "+v"(A(n)));
        REP8(X)
#undef X
      } else if (OP == LSHRREV_B32) {
#define X(n) asm volatile("v_lshrrev_b32 %0, 2, %0" : "+v"(A(n)));
        REP8(X)
#undef X
      } else if (OP == LSHL_ADD_U32) {
#define X(n) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(A(n)) : "v"(b));
        REP8(X)
#undef X
      } else if (OP == LSHL_OR_B32) {
#define X(n) asm volatile("v_lshl_or_b32 %0, %0, 2, %1" : "+v"(A(n)) : "v"(b));
        REP8(X)
#undef X
      } else if (OP == ADD3_U32) {
#define X(n) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(A(n)) : "v"(b), "v"(c));
        REP8(X)
#undef X
      } else if (OP == AND_OR_B32) {
#define X(n) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(A(n)) : "v"(b), "v"(c));
        REP8(X)
#undef X
      } else if (OP == XAD_U32) {
#define X(n) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(A(n)) : "v"(b), "v"(c));
        REP8(X)
#undef X
      } else if (OP == BFE_U32) {
#define X(n) asm volatile("v_bfe_u32 %0, %0, 2, 30" : "+v"(A(n)));
        REP8(X)
#undef X
      } else if (OP == BFI_B32) {
#define X(n) asm volatile("v_bfi_b32 %0, -4, %0, %1" : "+v"(A(n)) : "v"(b));
        REP8(X)
#undef X
      } else if (OP == ALIGNBIT_B32) {
#define X(n) asm volatile("v_alignbit_b32 %0, %0, %1, 30" : "+v"(A(n)) : "v"(b));
        REP8(X)
#undef X
      } else if (OP == ALIGNBYTE_B32) {
#define X(n) asm volatile("v_alignbyte_b32 %0, %0, %1, %2" : "+v"(A(n)) : "v"(b), "v"(c));
        REP8(X)
#undef X
      } else if (OP == PERM_B32) {
#define X(n) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(A(n)) : "v"(b), "v"(c));
        REP8(X)
#undef X
      } else if (OP == CNDMASK_VCC) {
        asm volatile("v_cmp_lt_u32 vcc, %0, %1" ::"v"(b), "v"(c) : "vcc");
#define X(n) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(A(n)) : "v"(b) : "vcc");
        REP8(X)
#undef X
      } else if (OP == CNDMASK_SGPR) {
        uint64_t m;
        asm volatile("v_cmp_lt_u32 %0, %1, %2" : "=s"(m) : "v"(b), "v"(c));
#define X(n) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(A(n)) : "v"(b), "s"(m));
        REP8(X)
#undef X
      } else if (OP == CMP_LT_U32) {
#define X(n) asm volatile("v_cmp_lt_u32 vcc, %0, %1" ::"v"(A(n)), "v"(b) : "vcc");
        REP8(X)
#undef X
      } else if (OP == CMP_LT_U32_SGPR) {
#define X(n) { uint64_t m; asm volatile("v_cmp_lt_u32 %0, %1, %2" : "=s"(m) : "v"(A(n)), "v"(b)); asm volatile("" ::"s"(m)); }
        REP8(X)
#undef X
      } else if (OP == CMP_LT_U64) {
#define X(n) asm volatile("v_cmp_lt_u64 vcc, %0, %1" ::"v"(W(n)), "v"(w0) : "vcc");
        REP8(X)
#undef X
      } else if (OP == LSHLREV_B64) {
#define X(n) asm volatile("v_lshlrev_b64 %0, 2, %0" : "+v"(W(n)));
        REP8(X)
#undef X
      } else if (OP == LSHRREV_B64) {
#define X(n) asm volatile("v_lshrrev_b64 %0, 2, %0" : "+v"(W(n)));
        REP8(X)
#undef X
      } else if (OP == MUL_LO_U32) {
#define X(n) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(A(n)) : "s"(s0));
        REP8(X)
#undef X
      } else if (OP == MUL_LO_U32_LIT) {
#define X(n) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(A(n)) : "v"(b));
        REP8(X)
#undef X
      } else if (OP == MUL_HI_U32) {
#define X(n) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(A(n)) : "s"(s0));
        REP8(X)
#undef X
      } else if (OP == MAD_U64_U32) {
#define X(n) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(W(n)) : "v"(A(n)), "s"(s0) : "vcc");
        REP8(X)
#undef X
      } else if (OP == MAD_U64_U32_ACC) {
#define X(n) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(W(n)) : "v"(A(n)), "s"(s0) : "vcc");
        REP8(X)
#undef X
      } else if (OP == MUL_U32_U24) {
#define X(n) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(A(n)) : "v"(b));
        REP8(X)
#undef X
      } else if (OP == MUL_HI_U32_U24) {
#define X(n) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(A(n)) : "v"(b));
        REP8(X)
#undef X
      } else if (OP == MAD_U32_U24) {
#define X(n) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(A(n)) : "v"(b), "v"(c));
        REP8(X)
#undef X
      } else if (OP == MAD_U32_U16) {
#define X(n) asm volatile("v_mad_u32_u16 %0, %0, %1, %2" : "+v"(A(n)) : "v"(b), "v"(c));
        REP8(X)
#undef X
      } else if (OP == PK_MUL_LO_U16) {
#define X(n) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(A(n)) : "v"(b));
        REP8(X)
#undef X
      } else if (OP == PK_ADD_U16) {
#define X(n) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(A(n)) : "v"(b));
        REP8(X)
#undef X
      } else if (OP == ADD_LSHL_U32) {
#define X(n) asm volatile("v_add_lshl_u32 %0, %0, %1, 3" : "+v"(A(n)) : "v"(b));
        REP8(X)
#undef X
      } else if (OP == SAD_U32) {
#define X(n) asm volatile("v_sad_u32 %0, %0, %1, %2" : "+v"(A(n)) : "v"(b), "v"(c));
        REP8(X)
#undef X
      } else if (OP == MBCNT_LO) {
#define X(n) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(A(n)) : "s"(s0));
        REP8(X)
#undef X
      } else if (OP == MBCNT_HI) {
#define X(n) asm volatile("v_mbcnt_hi_u32_b32 %0, %1, %0" : "+v"(A(n)) : "s"(s0));
        REP8(X)
#undef X
      } else if (OP == FFBH_U32) {
#define X(n) asm volatile("v_ffbh_u32 %0, %0" : "+v"(A(n)));
        REP8(X)
#undef X
      } else if (OP == MIN3_U32) {
#define X(n) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(A(n)) : "v"(b), "v"(c));
        REP8(X)
#undef X
      } else if (OP == MIN_U32) {
#define X(n) asm volatile("v_min_u32 %0, %0, %1" : "+v"(A(n)) : "v"(b));
        REP8(X)
#undef X
      } else if (OP == ADD_U32_SDWA) {
#define X(n) asm volatile("v_add_u32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(A(n)) : "v"(b));
        REP8(X)
#undef X
      } else if (OP == ADD_CO_U32) {
#define X(n) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(A(n)) : "v"(b) : "vcc");
        REP8(X)
#undef X
      } else if (OP == ADDC_CO_U32) {
#define X(n) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(A(n)) : "v"(b) : "vcc");
        REP8(X)
#undef X
      } else if (OP == SUB_CLAMP) {
#define X(n) asm volatile("v_sub_u32_e64 %0, %1, %0 clamp" : "+v"(A(n)) : "v"(b));
        REP8(X)
#undef X
      } else if (OP == READFIRSTLANE) {
#define X(n) { uint32_t s; asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(s) : "v"(A(n))); asm volatile("" ::"s"(s)); }
        REP8(X)
#undef X
      } else if (OP == READLANE) {
#define X(n) { uint32_t s; asm volatile("v_readlane_b32 %0, %1, 17" : "=s"(s) : "v"(A(n))); asm volatile("" ::"s"(s)); }
        REP8(X)
#undef X
      } else if (OP == MOV_DPP) {
#define X(n) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(A(n)));
        REP8(X)
#undef X
      } else if (OP == DS_WRITE_B64) {
#define X(n) asm volatile("ds_write_b64 %0, %1" ::"v"(ldsaddr), "v"(W(n)) : "memory");
        REP8(X)
#undef X
      } else if (OP == DS_READ_B64) {
#define X(n) asm volatile("ds_read_b64 %0, %1" : "=v"(W(n)) : "v"(ldsaddr) : "memory");
        REP8(X)
#undef X
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      } else if (OP == DS_READ_U8) {
#define X(n) asm volatile("ds_read_u8 %0, %1" : "=v"(A(n)) : "v"(c & 0xFFu) : "memory");
        REP8(X)
#undef X
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      } else if (OP == DS_MIN_U32_RANDOM) {
        // random cell among 4096 words, fresh per instruction (a sketch's ds_min pattern)
#define X(n) { uint32_t ad = ((A(n) * 2654435761u) >> 20) << 2; asm volatile("ds_min_u32 %0, %1" ::"v"(ad), "v"(b) : "memory"); A(n) += 0x9E3779B9u; }
        REP8(X)
#undef X
      } else if (OP == NOP) {
#define X(n) asm volatile("s_nop 0");
        REP8(X)
#undef X
      }
    }
  }
  if (threadIdx.x == 0) {
    uint64_t t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    if (blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
  }
  uint32_t x = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (uint32_t)(w0 ^ w1 ^ w2 ^ w3 ^ w4 ^ w5 ^ w6 ^ w7) ^ (uint32_t)((w0 ^ w1 ^ w2 ^ w3 ^ w4 ^ w5 ^ w6 ^ w7) >> 32);
  x ^= (uint32_t)lds64[(threadIdx.x * 7) & 2047];
  if (x == 0x12345u) out[0] = x;
}

static double g_add_cost = 0;

template <int OP>
void run(const char *name, int extra_valu_per_inst, uint32_t *out, uint64_t *clk, FILE *csv) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  const int blocks = 256, iters = 4000;   // one 1024-thread workgroup per CU: 4 waves per SIMD
  k<OP><<<blocks, 1024>>>(out, 50, 1, clk);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(a);
  k<OP><<<blocks, 1024>>>(out, iters, 1, clk);
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  uint64_t h[2]; (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  const double ghz = (double)h[0] / ((double)h[1] * 10.0);   // s_memtime ticks per ns (wall clock: 100 MHz)
  // wave-instructions per SIMD: 4 waves x iters x 64
  const double winst = 4.0 * iters * 64.0;
  const double cyc = (double)h[0] / winst;   // SIMD cycles per wave-instruction (by the kernel's own cycle counter)
  const double cyc_ev = ms * 1e6 * ghz / winst;
  double own = cyc;
  if (extra_valu_per_inst) own = cyc - extra_valu_per_inst * g_add_cost;
  (void)own;
  printf("%-22s %8.3f ms  %5.2f GHz  %6.2f SIMD cycles per wave-instruction (oldest wave alone: %5.2f)%s\n", name, ms, ghz, cyc_ev, cyc,
         extra_valu_per_inst ? "  [incl. address arithmetic]" : "");
  if (csv) fprintf(csv, "%s,%.4f,%.3f,%.3f,%.3f\n", name, ms, ghz, cyc_ev, cyc);
  if (OP == ADD_U32) g_add_cost = cyc_ev;
}

int main(int argc, char **argv) {
  uint32_t *out; uint64_t *clk;
  (void)hipMalloc(&out, 4096);
  (void)hipMalloc(&clk, 64);
  FILE *csv = argc > 1 ? fopen(argv[1], "w") : nullptr;
  if (csv) fprintf(csv, "opcode,ms,ghz,simd_cycles,oldest_wave_cycles\n");
  printf("# gfx950 opcode issue cost: 256 workgroups x 1024 threads (4 waves / SIMD), 8 independent chains per lane\n");
#define R(op, extra) run<op>(#op, extra, out, clk, csv)
  R(ADD_U32, 0); R(XOR_B32, 0); R(AND_B32, 0); R(OR_B32, 0); R(MOV_B32, 0); R(LSHLREV_B32, 0); R(LSHRREV_B32, 0);
  R(LSHL_ADD_U32, 0); R(LSHL_OR_B32, 0); R(ADD3_U32, 0); R(AND_OR_B32, 0); R(XAD_U32, 0); R(ADD_LSHL_U32, 0); R(SAD_U32, 0);
  R(BFE_U32, 0); R(BFI_B32, 0);
  R(ALIGNBIT_B32, 0); R(ALIGNBYTE_B32, 0); R(PERM_B32, 0); R(CNDMASK_VCC, 0); R(CNDMASK_SGPR, 0); R(CMP_LT_U32, 0); R(CMP_LT_U32_SGPR, 0);
  R(CMP_LT_U64, 0); R(LSHLREV_B64, 0); R(LSHRREV_B64, 0); R(MUL_LO_U32, 0); R(MUL_LO_U32_LIT, 0); R(MUL_HI_U32, 0);
  R(MAD_U64_U32, 0); R(MAD_U64_U32_ACC, 0); R(MUL_U32_U24, 0); R(MUL_HI_U32_U24, 0); R(MAD_U32_U24, 0); R(MAD_U32_U16, 0);
  R(PK_MUL_LO_U16, 0); R(PK_ADD_U16, 0);
  R(MBCNT_LO, 0); R(MBCNT_HI, 0); R(FFBH_U32, 0); R(MIN3_U32, 0); R(MIN_U32, 0); R(ADD_U32_SDWA, 0); R(ADD_CO_U32, 0);
  R(ADDC_CO_U32, 0); R(SUB_CLAMP, 0); R(READFIRSTLANE, 0); R(READLANE, 0); R(MOV_DPP, 0);
  R(DS_WRITE_B64, 0); R(DS_READ_B64, 0); R(DS_READ_U8, 1); R(DS_MIN_U32_RANDOM, 3); R(NOP, 0);
  if (csv) fclose(csv);
  return 0;
}
