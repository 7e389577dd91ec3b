// Issue-cost microbenchmark for gfx950: cycles per wave64 instruction per SIMD for the
// instruction kinds the march kernel is made of, at 1 / 2 / 4 / 8 waves per SIMD.
// Build: hipcc -O2 --offload-arch=gfx950 -o valu_rates valu_rates.hip ; run: ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kIters = 20000;
constexpr int kUnroll = 32;   // instructions per loop trip (8 independent chains x 4)

// 8 independent destination registers per kind so that consecutive instructions never depend
#define REP8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define REP32(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP)

#define KERNEL(NAME, DECL, BODY, SINK)                                                    \
  __global__ __launch_bounds__(256) void NAME(unsigned* out, const unsigned char* gbuf,    \
                                              unsigned long long* clocks) {                \
    __shared__ float4 lds[1024];                                                           \
    for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = make_float4(i, 1, 2, 3);        \
    __syncthreads();                                                                       \
    DECL                                                                                   \
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();                        \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();                            \
    for (int it = 0; it < kIters; ++it) {                                                  \
      BODY                                                                                 \
    }                                                                                      \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();                            \
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                        \
    if (threadIdx.x == 0) { clocks[2 * blockIdx.x] = t1 - t0; clocks[2 * blockIdx.x + 1] = r1 - r0; } \
    SINK                                                                                   \
  }

#define FDECL float a[8], b = threadIdx.x * 1e-3f + 1.0f, c = 0.5f; unsigned long long m[8]; \
  const unsigned long long mask64 = 0x5555555555555555ull ^ blockIdx.x; (void)mask64; (void)m; \
  _Pragma("unroll") for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;
#define FSINK float s = 0; _Pragma("unroll") for (int i = 0; i < 8; ++i) s += a[i]; \
  if (s == 12345.678f) out[0] = 1;
#define UDECL unsigned a[8], b = threadIdx.x * 7u + 3u, c = 5u; \
  _Pragma("unroll") for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;
#define USINK unsigned s = 0; _Pragma("unroll") for (int i = 0; i < 8; ++i) s += a[i]; \
  if (s == 0x12345678u) out[0] = 1;

#define OP_FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define OP_MUL(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_ADD(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_CVT(i) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a[i]));
#define OP_CVTU(i) asm volatile("v_cvt_u32_f32 %0, %0" : "+v"(a[i]));
#define OP_FLOOR(i) asm volatile("v_floor_f32 %0, %0" : "+v"(a[i]));
#define OP_MED3(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define OP_MIN(i) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
#define OP_CMP(i) asm volatile("v_cmp_lt_f32 %0, %1, %2" : "=s"(m[i]) : "v"(a[i]), "v"(b));
#define OP_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "s"(mask64));
#define OP_SHR(i) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(a[i]));
#define OP_SHL(i) asm volatile("v_lshlrev_b32 %0, 2, %0" : "+v"(a[i]));
#define OP_LSHLADD(i) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a[i]) : "v"(b));
#define OP_MADU24(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define OP_ADD3(i) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define OP_AND(i) asm volatile("v_and_b32 %0, 0xff, %0" : "+v"(a[i]));
#define OP_BFE(i) asm volatile("v_bfe_u32 %0, %0, 3, 5" : "+v"(a[i]));
#define OP_ANDOR(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define OP_LSHLOR(i) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(a[i]) : "v"(b));
#define OP_MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_ADDU(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_PERM(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define OP_MOV(i) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b));
#define OP_NOP(i) asm volatile("s_nop 0");

KERNEL(k_fma, FDECL, REP32(OP_FMA), FSINK)
KERNEL(k_mul, FDECL, REP32(OP_MUL), FSINK)
KERNEL(k_add, FDECL, REP32(OP_ADD), FSINK)
KERNEL(k_cvt, FDECL, REP32(OP_CVT), FSINK)
KERNEL(k_cvtu, FDECL, REP32(OP_CVTU), FSINK)
KERNEL(k_floor, FDECL, REP32(OP_FLOOR), FSINK)
KERNEL(k_med3, FDECL, REP32(OP_MED3), FSINK)
KERNEL(k_min, FDECL, REP32(OP_MIN), FSINK)
KERNEL(k_rcp, FDECL, REP32(OP_RCP), FSINK)
KERNEL(k_cmp, FDECL, REP32(OP_CMP), FSINK)
KERNEL(k_cndmask, FDECL, REP32(OP_CNDMASK), FSINK)
KERNEL(k_shr, UDECL, REP32(OP_SHR), USINK)
KERNEL(k_shl, UDECL, REP32(OP_SHL), USINK)
KERNEL(k_lshladd, UDECL, REP32(OP_LSHLADD), USINK)
KERNEL(k_madu24, UDECL, REP32(OP_MADU24), USINK)
KERNEL(k_add3, UDECL, REP32(OP_ADD3), USINK)
KERNEL(k_and, UDECL, REP32(OP_AND), USINK)
KERNEL(k_bfe, UDECL, REP32(OP_BFE), USINK)
KERNEL(k_andor, UDECL, REP32(OP_ANDOR), USINK)
KERNEL(k_lshlor, UDECL, REP32(OP_LSHLOR), USINK)
KERNEL(k_mullo, UDECL, REP32(OP_MULLO), USINK)
KERNEL(k_addu, UDECL, REP32(OP_ADDU), USINK)
KERNEL(k_perm, UDECL, REP32(OP_PERM), USINK)
KERNEL(k_mov, UDECL, REP32(OP_MOV), USINK)
KERNEL(k_snop, UDECL, REP32(OP_NOP), USINK)

// packed f32: 8 independent register pairs
typedef float float2_t __attribute__((ext_vector_type(2)));
#define PDECL float2_t a[8], b = {threadIdx.x * 1e-3f + 1.0f, 1.5f}, c = {0.5f, 0.25f}; \
  _Pragma("unroll") for (int i = 0; i < 8; ++i) a[i] = float2_t{(float)threadIdx.x, (float)i};
#define PSINK float s = 0; _Pragma("unroll") for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y; \
  if (s == 12345.678f) out[0] = 1;
#define OP_PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_PKADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
KERNEL(k_pkmul, PDECL, REP32(OP_PKMUL), PSINK)
KERNEL(k_pkadd, PDECL, REP32(OP_PKADD), PSINK)
KERNEL(k_pkfma, PDECL, REP32(OP_PKFMA), PSINK)

// LDS reads: address pattern chosen by MODE (0 = all lanes same address, 1 = lane-linear,
// 2 = pseudo-random 16-byte entries as the transfer-function lookup does)
#define LDECL unsigned a[8]; float4 v[8]; \
  _Pragma("unroll") for (int i = 0; i < 8; ++i) { \
    unsigned lane = threadIdx.x & 63; \
    unsigned idx = (MODE == 0) ? i : (MODE == 1) ? (lane + i) : ((lane * 2654435761u >> 8) + i * 37u); \
    a[i] = (idx & 255u) * 16u; v[i] = make_float4(0, 0, 0, 0); }
#define LSINK float s = 0; _Pragma("unroll") for (int i = 0; i < 8; ++i) s += v[i].x + v[i].w; \
  if (s == 12345.678f) out[0] = 1;
#define OP_DSB128(i) asm volatile("ds_read_b128 %0, %1" : "=v"(v[i]) : "v"(a[i]));
#define OP_DSB32(i) asm volatile("ds_read_b32 %0, %1" : "=v"(v[i].x) : "v"(a[i]));
#define OP_DSU8(i) asm volatile("ds_read_u8 %0, %1" : "=v"(v[i].x) : "v"(a[i]));
#define LWAIT asm volatile("s_waitcnt lgkmcnt(0)");
template <int MODE> KERNEL(k_ds128, LDECL, REP8(OP_DSB128) LWAIT REP8(OP_DSB128) LWAIT REP8(OP_DSB128) LWAIT REP8(OP_DSB128) LWAIT, LSINK)
template <int MODE> KERNEL(k_ds32, LDECL, REP8(OP_DSB32) LWAIT REP8(OP_DSB32) LWAIT REP8(OP_DSB32) LWAIT REP8(OP_DSB32) LWAIT, LSINK)
template <int MODE> KERNEL(k_dsu8, LDECL, REP8(OP_DSU8) LWAIT REP8(OP_DSU8) LWAIT REP8(OP_DSU8) LWAIT REP8(OP_DSU8) LWAIT, LSINK)

// byte gathers from a small (L1-resident) buffer: LINES distinct 128-byte lines per wave instruction
#define GDECL unsigned off[8]; unsigned v[8]; \
  _Pragma("unroll") for (int i = 0; i < 8; ++i) { \
    unsigned lane = threadIdx.x & 63; \
    off[i] = ((lane % LINES) * 128u + (lane / LINES) + i * 2048u) & 16383u; v[i] = 0; }
#define GSINK unsigned s = 0; _Pragma("unroll") for (int i = 0; i < 8; ++i) s += v[i]; \
  if (s == 0x12345678u) out[0] = 1;
#define OP_GLDU8(i) asm volatile("global_load_ubyte %0, %1, %2" : "=v"(v[i]) : "v"(off[i]), "s"(gbuf));
#define GWAIT asm volatile("s_waitcnt vmcnt(0)");
template <int LINES> KERNEL(k_gld, GDECL, REP8(OP_GLDU8) GWAIT REP8(OP_GLDU8) GWAIT REP8(OP_GLDU8) GWAIT REP8(OP_GLDU8) GWAIT, GSINK)

// The instruction multiset of one interior-loop sample of the round-1 march (no memory ops):
// 5 packed f32, 3 cvt, 3 shifts, 2 lshl_add, 3 mad_u24, 1 shl (table index), 7 scalar f32
#define OP_MIX(i) asm volatile( \
    "v_pk_mul_f32 %0, %0, %2\n v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %0, %0, %2\n" \
    "v_pk_mul_f32 %0, %0, %2\n v_pk_mul_f32 %0, %0, %2\n" \
    "v_cvt_i32_f32 %1, %1\n v_cvt_i32_f32 %1, %1\n v_cvt_i32_f32 %1, %1\n" \
    "v_lshrrev_b32 %1, 3, %1\n v_lshrrev_b32 %1, 2, %1\n v_lshrrev_b32 %1, 2, %1\n" \
    "v_lshl_add_u32 %1, %1, 3, %1\n v_lshl_add_u32 %1, %1, 5, %1\n" \
    "v_mad_u32_u24 %1, %1, %1, %1\n v_mad_u32_u24 %1, %1, %1, %1\n v_mad_u32_u24 %1, %1, %1, %1\n" \
    "v_lshlrev_b32 %1, 4, %1\n" \
    "v_sub_f32 %1, %1, %1\n v_mul_f32 %1, %1, %1\n v_add_f32 %1, %1, %1\n" \
    "v_mul_f32 %1, %1, %1\n v_add_f32 %1, %1, %1\n v_mul_f32 %1, %1, %1\n v_add_f32 %1, %1, %1\n" \
    : "+v"(a[i]), "+v"(u[i]) : "v"(b));
#define MDECL float2_t a[8], b = {1.0f, 1.5f}; float u[8]; \
  _Pragma("unroll") for (int i = 0; i < 8; ++i) { a[i] = float2_t{(float)threadIdx.x, (float)i}; u[i] = i; }
#define MSINK float s = 0; _Pragma("unroll") for (int i = 0; i < 8; ++i) s += a[i].x + u[i]; \
  if (s == 12345.678f) out[0] = 1;
KERNEL(k_mix, MDECL, REP32(OP_MIX), MSINK)

#define DDECL double a[8], b = threadIdx.x * 1e-3 + 1.0, c = 0.5; float u[8]; unsigned long long m[8]; \
  const unsigned cls = 0x1f8; (void)cls; (void)m; (void)c; \
  _Pragma("unroll") for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x + i; u[i] = 0; }
#define DSINK double s = 0; _Pragma("unroll") for (int i = 0; i < 8; ++i) s += a[i] + u[i]; \
  if (s == 12345.678) out[0] = 1;
#define OP_X_subf(i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_X_maxf(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_X_trunc(i) asm volatile("v_trunc_f32 %0, %0" : "+v"(a[i]));
#define OP_X_fract(i) asm volatile("v_fract_f32 %0, %0" : "+v"(a[i]));
#define OP_X_rndne(i) asm volatile("v_rndne_f32 %0, %0" : "+v"(a[i]));
#define OP_X_fmac(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define OP_X_mulleg(i) asm volatile("v_mul_legacy_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_X_ldexp(i) asm volatile("v_ldexp_f32 %0, %0, 3" : "+v"(a[i]));
#define OP_X_cvtf32u32(i) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a[i]));
#define OP_X_cvtf32ub0(i) asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(a[i]));
#define OP_X_max3f(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define OP_X_or(i) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_X_xor(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_X_subu(i) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_X_subrevu(i) asm volatile("v_subrev_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_X_ashr(i) asm volatile("v_ashrrev_i32 %0, 3, %0" : "+v"(a[i]));
#define OP_X_mulu24(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_X_muli24(i) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_X_madi24(i) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define OP_X_mulhiu24(i) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_X_mulhi(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_X_bfi(i) asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define OP_X_alignbit(i) asm volatile("v_alignbit_b32 %0, %0, %1, 3" : "+v"(a[i]) : "v"(b));
#define OP_X_alignbyte(i) asm volatile("v_alignbyte_b32 %0, %0, %1, 1" : "+v"(a[i]) : "v"(b));
#define OP_X_dot4(i) asm volatile("v_dot4_u32_u8 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define OP_X_sad(i) asm volatile("v_sad_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define OP_X_cvtpku8(i) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(a[i]) : "v"(b));
#define OP_X_minu(i) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_X_maxu(i) asm volatile("v_max_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_X_med3u(i) asm volatile("v_med3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define OP_X_xad(i) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define OP_X_addlshl(i) asm volatile("v_add_lshl_u32 %0, %0, %1, 3" : "+v"(a[i]) : "v"(b));
#define OP_X_or3(i) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define OP_X_lshrrev_sdwa(i) asm volatile("v_lshrrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "+v"(a[i]) : "v"(c));
#define OP_X_mov_sdwa(i) asm volatile("v_mov_b32_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1" : "+v"(a[i]));
#define OP_X_and_sdwa(i) asm volatile("v_and_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(a[i]) : "v"(b));
#define OP_X_add_sdwa(i) asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "+v"(a[i]) : "v"(b));
#define OP_X_mov_dpp(i) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
#define OP_X_addu_dpp(i) asm volatile("v_add_u32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b));
#define OP_X_addu_lit(i) asm volatile("v_add_u32 %0, 0x12345, %0" : "+v"(a[i]));
#define OP_X_lshl_lit1(i) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a[i]));
#define OP_X_addf64(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_X_mulf64(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_X_fmaf64(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define OP_X_cvtf32f64(i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(u[i]) : "v"(a[i]));
#define OP_X_cmpclassf64(i) asm volatile("v_cmp_class_f64 %0, %1, %2" : "=s"(m[i]) : "v"(a[i]), "v"(cls));
KERNEL(kx_subf, FDECL, REP32(OP_X_subf), FSINK)
KERNEL(kx_maxf, FDECL, REP32(OP_X_maxf), FSINK)
KERNEL(kx_trunc, FDECL, REP32(OP_X_trunc), FSINK)
KERNEL(kx_fract, FDECL, REP32(OP_X_fract), FSINK)
KERNEL(kx_rndne, FDECL, REP32(OP_X_rndne), FSINK)
KERNEL(kx_fmac, FDECL, REP32(OP_X_fmac), FSINK)
KERNEL(kx_mulleg, FDECL, REP32(OP_X_mulleg), FSINK)
KERNEL(kx_ldexp, FDECL, REP32(OP_X_ldexp), FSINK)
KERNEL(kx_cvtf32u32, FDECL, REP32(OP_X_cvtf32u32), FSINK)
KERNEL(kx_cvtf32ub0, FDECL, REP32(OP_X_cvtf32ub0), FSINK)
KERNEL(kx_max3f, FDECL, REP32(OP_X_max3f), FSINK)
KERNEL(kx_or, UDECL, REP32(OP_X_or), USINK)
KERNEL(kx_xor, UDECL, REP32(OP_X_xor), USINK)
KERNEL(kx_subu, UDECL, REP32(OP_X_subu), USINK)
KERNEL(kx_subrevu, UDECL, REP32(OP_X_subrevu), USINK)
KERNEL(kx_ashr, UDECL, REP32(OP_X_ashr), USINK)
KERNEL(kx_mulu24, UDECL, REP32(OP_X_mulu24), USINK)
KERNEL(kx_muli24, UDECL, REP32(OP_X_muli24), USINK)
KERNEL(kx_madi24, UDECL, REP32(OP_X_madi24), USINK)
KERNEL(kx_mulhiu24, UDECL, REP32(OP_X_mulhiu24), USINK)
KERNEL(kx_mulhi, UDECL, REP32(OP_X_mulhi), USINK)
KERNEL(kx_bfi, UDECL, REP32(OP_X_bfi), USINK)
KERNEL(kx_alignbit, UDECL, REP32(OP_X_alignbit), USINK)
KERNEL(kx_alignbyte, UDECL, REP32(OP_X_alignbyte), USINK)
KERNEL(kx_dot4, UDECL, REP32(OP_X_dot4), USINK)
KERNEL(kx_sad, UDECL, REP32(OP_X_sad), USINK)
KERNEL(kx_cvtpku8, UDECL, REP32(OP_X_cvtpku8), USINK)
KERNEL(kx_minu, UDECL, REP32(OP_X_minu), USINK)
KERNEL(kx_maxu, UDECL, REP32(OP_X_maxu), USINK)
KERNEL(kx_med3u, UDECL, REP32(OP_X_med3u), USINK)
KERNEL(kx_xad, UDECL, REP32(OP_X_xad), USINK)
KERNEL(kx_addlshl, UDECL, REP32(OP_X_addlshl), USINK)
KERNEL(kx_or3, UDECL, REP32(OP_X_or3), USINK)
KERNEL(kx_lshrrev_sdwa, UDECL, REP32(OP_X_lshrrev_sdwa), USINK)
KERNEL(kx_mov_sdwa, UDECL, REP32(OP_X_mov_sdwa), USINK)
KERNEL(kx_and_sdwa, UDECL, REP32(OP_X_and_sdwa), USINK)
KERNEL(kx_add_sdwa, UDECL, REP32(OP_X_add_sdwa), USINK)
KERNEL(kx_mov_dpp, UDECL, REP32(OP_X_mov_dpp), USINK)
KERNEL(kx_addu_dpp, UDECL, REP32(OP_X_addu_dpp), USINK)
KERNEL(kx_addu_lit, UDECL, REP32(OP_X_addu_lit), USINK)
KERNEL(kx_lshl_lit1, UDECL, REP32(OP_X_lshl_lit1), USINK)
KERNEL(kx_addf64, DDECL, REP32(OP_X_addf64), DSINK)
KERNEL(kx_mulf64, DDECL, REP32(OP_X_mulf64), DSINK)
KERNEL(kx_fmaf64, DDECL, REP32(OP_X_fmaf64), DSINK)
KERNEL(kx_cvtf32f64, DDECL, REP32(OP_X_cvtf32f64), DSINK)
KERNEL(kx_cmpclassf64, DDECL, REP32(OP_X_cmpclassf64), DSINK)

#define REP32_1(OP) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) \
                    OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0) OP(0)
#define REP32_2(OP) OP(0) OP(1) OP(0) OP(1) OP(0) OP(1) OP(0) OP(1) OP(0) OP(1) OP(0) OP(1) OP(0) OP(1) OP(0) OP(1) \
                    OP(0) OP(1) OP(0) OP(1) OP(0) OP(1) OP(0) OP(1) OP(0) OP(1) OP(0) OP(1) OP(0) OP(1) OP(0) OP(1)
#define REP32_4(OP) REP8_4(OP) REP8_4(OP) REP8_4(OP) REP8_4(OP)
#define REP8_4(OP) OP(0) OP(1) OP(2) OP(3) OP(0) OP(1) OP(2) OP(3)
KERNEL(k_add_dep1, FDECL, REP32_1(OP_ADD), FSINK)
KERNEL(k_add_dep2, FDECL, REP32_2(OP_ADD), FSINK)
KERNEL(k_add_dep4, FDECL, REP32_4(OP_ADD), FSINK)
KERNEL(k_fma_dep1, FDECL, REP32_1(OP_FMA), FSINK)
KERNEL(k_cvt_dep1, FDECL, REP32_1(OP_CVT), FSINK)
KERNEL(k_madu24_dep1, UDECL, REP32_1(OP_MADU24), USINK)
KERNEL(k_madu24_dep2, UDECL, REP32_2(OP_MADU24), USINK)
// alternating fast / slow, dependent
#define OP_ALT(i) asm volatile("v_add_f32 %0, %0, %1\n v_cvt_i32_f32 %0, %0" : "+v"(a[i]) : "v"(b));
KERNEL(k_alt_dep1, FDECL, REP32_1(OP_ALT), FSINK)
KERNEL(k_alt_dep4, FDECL, REP32_4(OP_ALT), FSINK)

// large loop bodies: does instruction fetch limit issue once the loop no longer fits the wave's
// instruction buffer?  (kUnroll is 32: run kIters/32 trips of 1024 instructions)
#define REP1024(OP) REP32(OP) REP32(OP) REP32(OP) REP32(OP) REP32(OP) REP32(OP) REP32(OP) REP32(OP) \
  REP32(OP) REP32(OP) REP32(OP) REP32(OP) REP32(OP) REP32(OP) REP32(OP) REP32(OP) \
  REP32(OP) REP32(OP) REP32(OP) REP32(OP) REP32(OP) REP32(OP) REP32(OP) REP32(OP) \
  REP32(OP) REP32(OP) REP32(OP) REP32(OP) REP32(OP) REP32(OP) REP32(OP) REP32(OP)
#define KERNEL_BIG(NAME, DECL, BODY, SINK)                                                \
  __global__ __launch_bounds__(256) void NAME(unsigned* out, const unsigned char* gbuf,    \
                                              unsigned long long* clocks) {                \
    __shared__ float4 lds[1024];                                                           \
    for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = make_float4(i, 1, 2, 3);        \
    __syncthreads();                                                                       \
    DECL                                                                                   \
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();                        \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();                            \
    for (int it = 0; it < kIters / 32; ++it) {                                             \
      BODY                                                                                 \
    }                                                                                      \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();                            \
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                        \
    if (threadIdx.x == 0) { clocks[2 * blockIdx.x] = t1 - t0; clocks[2 * blockIdx.x + 1] = r1 - r0; } \
    SINK                                                                                   \
  }
KERNEL_BIG(k_add_big, FDECL, REP1024(OP_ADD), FSINK)
KERNEL_BIG(k_fma_big, FDECL, REP1024(OP_FMA), FSINK)
KERNEL_BIG(k_madu24_big, UDECL, REP1024(OP_MADU24), USINK)

// operand kinds: SGPR, inline constant, 32-bit literal
#define SDECL float a[8]; float sb = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, 1.0f + blockIdx.x * 1e-3f))); \
  float vc = threadIdx.x * 0.5f; (void)vc; \
  _Pragma("unroll") for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;
#define OP_ADD_S(i) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "s"(sb));
#define OP_MUL_S(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "s"(sb));
#define OP_FMA_S(i) asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(a[i]) : "s"(sb), "v"(vc));
#define OP_ADD_K(i) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(a[i]));
#define OP_ADD_L(i) asm volatile("v_add_f32 %0, 0x3f8ccccd, %0" : "+v"(a[i]));
#define OP_SUB_K(i) asm volatile("v_sub_f32 %0, 1.0, %0" : "+v"(a[i]));
#define OP_LSHR_V(i) asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(a[i]) : "v"(vc));
#define OP_MADU24_S(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "s"(sb));
#define OP_ADD_E64(i) asm volatile("v_add_f32_e64 %0, %0, %1" : "+v"(a[i]) : "v"(vc));
#define OP_MUL_E64(i) asm volatile("v_mul_f32_e64 %0, %0, %1" : "+v"(a[i]) : "v"(vc));
KERNEL(k_add_s, SDECL, REP32(OP_ADD_S), FSINK)
KERNEL(k_mul_s, SDECL, REP32(OP_MUL_S), FSINK)
KERNEL(k_fma_s, SDECL, REP32(OP_FMA_S), FSINK)
KERNEL(k_add_k, SDECL, REP32(OP_ADD_K), FSINK)
KERNEL(k_add_l, SDECL, REP32(OP_ADD_L), FSINK)
KERNEL(k_sub_k, SDECL, REP32(OP_SUB_K), FSINK)
KERNEL(k_lshr_v, SDECL, REP32(OP_LSHR_V), FSINK)
KERNEL(k_madu24_s, SDECL, REP32(OP_MADU24_S), FSINK)
KERNEL(k_add_e64, SDECL, REP32(OP_ADD_E64), FSINK)
KERNEL(k_mul_e64, SDECL, REP32(OP_MUL_E64), FSINK)

struct Entry { const char* name; void (*fn)(unsigned*, const unsigned char*, unsigned long long*); };

int main() {
  unsigned* out; unsigned char* gbuf; unsigned long long* clocks;
  CHECK(hipMalloc(&out, 64));
  CHECK(hipMalloc(&gbuf, 1 << 20));
  CHECK(hipMemset(gbuf, 1, 1 << 20));
  CHECK(hipMalloc(&clocks, 16 * 2048));
  std::vector<Entry> entries = {
    {"v_fma_f32", k_fma}, {"v_mul_f32", k_mul}, {"v_add_f32", k_add}, {"v_cvt_i32_f32", k_cvt},
    {"v_cvt_u32_f32", k_cvtu}, {"v_floor_f32", k_floor}, {"v_med3_f32", k_med3}, {"v_min_f32", k_min},
    {"v_rcp_f32", k_rcp}, {"v_cmp_lt_f32", k_cmp}, {"v_cndmask_b32", k_cndmask},
    {"v_lshrrev_b32", k_shr}, {"v_lshlrev_b32", k_shl}, {"v_lshl_add_u32", k_lshladd},
    {"v_mad_u32_u24", k_madu24}, {"v_add3_u32", k_add3}, {"v_and_b32", k_and}, {"v_bfe_u32", k_bfe},
    {"v_and_or_b32", k_andor}, {"v_lshl_or_b32", k_lshlor}, {"v_mul_lo_u32", k_mullo},
    {"v_add_u32", k_addu}, {"v_perm_b32", k_perm}, {"v_mov_b32", k_mov}, {"s_nop", k_snop},
    {"v_pk_mul_f32", k_pkmul}, {"v_pk_add_f32", k_pkadd}, {"v_pk_fma_f32", k_pkfma},
    {"v_sub_f32 [subf]", kx_subf},
    {"v_max_f32 [maxf]", kx_maxf},
    {"v_trunc_f32 [trunc]", kx_trunc},
    {"v_fract_f32 [fract]", kx_fract},
    {"v_rndne_f32 [rndne]", kx_rndne},
    {"v_fmac_f32 [fmac]", kx_fmac},
    {"v_mul_legacy_f32 [mulleg]", kx_mulleg},
    {"v_ldexp_f32 [ldexp]", kx_ldexp},
    {"v_cvt_f32_u32 [cvtf32u32]", kx_cvtf32u32},
    {"v_cvt_f32_ubyte0 [cvtf32ub0]", kx_cvtf32ub0},
    {"v_max3_f32 [max3f]", kx_max3f},
    {"v_or_b32 [or]", kx_or},
    {"v_xor_b32 [xor]", kx_xor},
    {"v_sub_u32 [subu]", kx_subu},
    {"v_subrev_u32 [subrevu]", kx_subrevu},
    {"v_ashrrev_i32 [ashr]", kx_ashr},
    {"v_mul_u32_u24 [mulu24]", kx_mulu24},
    {"v_mul_i32_i24 [muli24]", kx_muli24},
    {"v_mad_i32_i24 [madi24]", kx_madi24},
    {"v_mul_hi_u32_u24 [mulhiu24]", kx_mulhiu24},
    {"v_mul_hi_u32 [mulhi]", kx_mulhi},
    {"v_bfi_b32 [bfi]", kx_bfi},
    {"v_alignbit_b32 [alignbit]", kx_alignbit},
    {"v_alignbyte_b32 [alignbyte]", kx_alignbyte},
    {"v_dot4_u32_u8 [dot4]", kx_dot4},
    {"v_sad_u32 [sad]", kx_sad},
    {"v_cvt_pk_u8_f32 [cvtpku8]", kx_cvtpku8},
    {"v_min_u32 [minu]", kx_minu},
    {"v_max_u32 [maxu]", kx_maxu},
    {"v_med3_u32 [med3u]", kx_med3u},
    {"v_xad_u32 [xad]", kx_xad},
    {"v_add_lshl_u32 [addlshl]", kx_addlshl},
    {"v_or3_b32 [or3]", kx_or3},
    {"v_lshrrev_b32_sdwa [lshrrev_sdwa]", kx_lshrrev_sdwa},
    {"v_mov_b32_sdwa [mov_sdwa]", kx_mov_sdwa},
    {"v_and_b32_sdwa [and_sdwa]", kx_and_sdwa},
    {"v_add_u32_sdwa [add_sdwa]", kx_add_sdwa},
    {"v_mov_b32_dpp [mov_dpp]", kx_mov_dpp},
    {"v_add_u32_dpp [addu_dpp]", kx_addu_dpp},
    {"v_add_u32 [addu_lit]", kx_addu_lit},
    {"v_lshlrev_b32 [lshl_lit1]", kx_lshl_lit1},
    {"v_add_f64 [addf64]", kx_addf64},
    {"v_mul_f64 [mulf64]", kx_mulf64},
    {"v_fma_f64 [fmaf64]", kx_fmaf64},
    {"v_cvt_f32_f64 [cvtf32f64]", kx_cvtf32f64},
    {"v_cmp_class_f64 [cmpclassf64]", kx_cmpclassf64},
    {"v_add_f32 1 chain", k_add_dep1}, {"v_add_f32 2 chains", k_add_dep2}, {"v_add_f32 4 chains", k_add_dep4},
    {"v_fma_f32 1 chain", k_fma_dep1}, {"v_cvt_i32_f32 1 chain", k_cvt_dep1},
    {"v_mad_u32_u24 1 chain", k_madu24_dep1}, {"v_mad_u32_u24 2 chains", k_madu24_dep2},
    {"add+cvt pair 1 chain /2", k_alt_dep1}, {"add+cvt pair 4 chains /2", k_alt_dep4},
    {"v_add_f32 1024-instr chain loop", k_add_big}, {"v_fma_f32 1024-instr chain loop", k_fma_big},
    {"v_mad_u32_u24 1024-instr chain loop", k_madu24_big},
    {"operand: v_add_f32 sgpr", k_add_s}, {"operand: v_mul_f32 sgpr", k_mul_s}, {"operand: v_fma_f32 sgpr", k_fma_s},
    {"operand: v_add_f32 inline 1.0", k_add_k}, {"operand: v_add_f32 literal", k_add_l},
    {"operand: v_sub_f32 inline 1.0", k_sub_k}, {"operand: v_lshrrev vgpr shift", k_lshr_v},
    {"operand: v_mad_u32_u24 sgpr", k_madu24_s}, {"operand: v_add_f32_e64", k_add_e64}, {"operand: v_mul_f32_e64", k_mul_e64},
    {"march mix (24 VALU) /24", k_mix},
    {"ds_read_b128 same", k_ds128<0>}, {"ds_read_b128 linear", k_ds128<1>}, {"ds_read_b128 random", k_ds128<2>},
    {"ds_read_b32 same", k_ds32<0>}, {"ds_read_b32 random", k_ds32<2>}, {"ds_read_u8 random", k_dsu8<2>},
    {"global_load_ubyte 1 line", k_gld<1>}, {"global_load_ubyte 2 lines", k_gld<2>},
    {"global_load_ubyte 4 lines", k_gld<4>}, {"global_load_ubyte 8 lines", k_gld<8>},
    {"global_load_ubyte 16 lines", k_gld<16>}, {"global_load_ubyte 64 lines", k_gld<64>},
  };
  printf("%-28s %10s %10s %10s %10s   (shader cycles per wave-instruction per SIMD; 256 CUs busy)\n",
         "instruction", "1 w/SIMD", "2 w/SIMD", "4 w/SIMD", "8 w/SIMD");
  std::vector<unsigned long long> host(4096);
  hipEvent_t ev0, ev1;
  CHECK(hipEventCreate(&ev0));
  CHECK(hipEventCreate(&ev1));
  const char* only = getenv("ONLY");
  for (const Entry& e : entries) {
    if (only && !strstr(e.name, only)) continue;
    printf("%-28s", e.name);
    for (int wps : {1, 2, 4, 8}) {
      const int blocks = 256 * wps;   // 256-thread blocks: one wave per SIMD each
      hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out, gbuf, clocks);
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(ev0, 0));
      hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out, gbuf, clocks);
      CHECK(hipEventRecord(ev1, 0));
      CHECK(hipDeviceSynchronize());
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, ev0, ev1));
      CHECK(hipMemcpy(host.data(), clocks, blocks * 16, hipMemcpyDeviceToHost));
      double ticks = 0, real = 0;
      for (int b = 0; b < blocks; ++b) { ticks += (double)host[2 * b]; real += (double)host[2 * b + 1]; }
      const double ghz = ticks / real * 0.1;            // s_memrealtime ticks at 100 MHz
      // wall-clock based: every SIMD issued wps * kIters * kUnroll wave-instructions
      const double cycles = (double)ms * 1e6 * ghz / ((double)kIters * kUnroll * wps);
      printf(" %10.2f", cycles);
      if (wps == 8) {
        int resident = 0;
        CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, e.fn, 256, 0));
        printf("   [8w: %.2f ms, %.2f GHz, %d blocks/CU resident, in-block %.2f]", ms, ghz, resident,
               ticks / blocks / ((double)kIters * kUnroll * wps));
      }
    }
    printf("\n");
    fflush(stdout);
  }
  return 0;
}
