// Micro-benchmark (round 5, second table): cycles per wave64 instruction and SIMD for the instruction classes the traversal turns are
// made of besides the ones valu_rate.hip measured -- plain fp32 add / mul / fma with input modifiers, min3 / max3, compares, selects,
// bit operations, integer add / shift / min / max, moves -- so that instruction SELECTION in the hot loops can be priced
// (e.g. v_and + half a v_pk_fma against one v_fma_f32 with an |x| modifier).  Same method as valu_rate.hip: 8 independent chains
// per lane, blocks of 64 instructions in inline assembly, 8 waves per SIMD (256 CUs x 8 blocks of 256 threads), events, nominal 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITERS 1500
#define A8(op, t) op " %0, %0" t "\n" op " %1, %1" t "\n" op " %2, %2" t "\n" op " %3, %3" t "\n" op " %4, %4" t "\n" op " %5, %5" t "\n" op " %6, %6" t "\n" op " %7, %7" t "\n"
#define A64(op, t) A8(op, t) A8(op, t) A8(op, t) A8(op, t) A8(op, t) A8(op, t) A8(op, t) A8(op, t)
// destination-only form (compares: the result goes to vcc)
#define C8(op, t) op " vcc, %0" t "\n" op " vcc, %1" t "\n" op " vcc, %2" t "\n" op " vcc, %3" t "\n" op " vcc, %4" t "\n" op " vcc, %5" t "\n" op " vcc, %6" t "\n" op " vcc, %7" t "\n"
#define C64(op, t) C8(op, t) C8(op, t) C8(op, t) C8(op, t) C8(op, t) C8(op, t) C8(op, t) C8(op, t)
#define KERNEL(name, T, BODY, ...)                                                                                           \
  __global__ __launch_bounds__(256) void name(float* out) {                                                                  \
    T a[8];                                                                                                                  \
    T b = (T)(threadIdx.x + 3), c = (T)(threadIdx.x * 5 + 1);                                                                \
    _Pragma("unroll") for (int k = 0; k < 8; k++) a[k] = (T)(threadIdx.x + k + 1);                                           \
    _Pragma("unroll 1") for (int i = 0; i < ITERS; ++i) {                                                                    \
      asm volatile(BODY : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(c) : __VA_ARGS__); \
    }                                                                                                                        \
    T s = 0;                                                                                                                 \
    for (int k = 0; k < 8; k++) s += a[k];                                                                                   \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)s;                                                                   \
  }
KERNEL(u_add_f32, float, A64("v_add_f32", ", %8"), "memory")
KERNEL(u_mul_f32, float, A64("v_mul_f32", ", %8"), "memory")
KERNEL(u_sub_f32, float, A64("v_sub_f32", ", %8"), "memory")
KERNEL(u_fma_abs, float, A64("v_fma_f32", ", -|%8|, %9"), "memory")
KERNEL(u_fmac, float, A64("v_fmac_f32", ", %8"), "memory")
KERNEL(u_max_f32, float, A64("v_max_f32", ", %8"), "memory")
KERNEL(u_max3_f32, float, A64("v_max3_f32", ", %8, %9"), "memory")
KERNEL(u_min3_f32, float, A64("v_min3_f32", ", %8, %9"), "memory")
KERNEL(u_cmp_le_f32, float, C64("v_cmp_le_f32", ", %8"), "memory", "vcc")
KERNEL(u_cndmask, unsigned, A64("v_cndmask_b32", ", %8, vcc"), "memory")
KERNEL(u_cndmask_sgpr, unsigned, "s_mov_b64 s[20:21], 0x5555\n" A64("v_cndmask_b32", ", %8, s[20:21]"), "memory", "s20", "s21")
KERNEL(u_cndmask_init, unsigned, "v_cmp_lt_u32 vcc, %8, %9\n" A64("v_cndmask_b32", ", %8, vcc"), "memory", "vcc")
KERNEL(u_cndmask_e64vcc, unsigned, A64("v_cndmask_b32_e64", ", %8, vcc"), "memory")
#define M4(c) "v_cndmask_b32 %0, %0, %8, " c "\nv_add_u32 %1, %1, %8\nv_add_u32 %2, %2, %8\nv_add_u32 %3, %3, %8\nv_cndmask_b32 %4, %4, %8, " c "\nv_add_u32 %5, %5, %8\nv_add_u32 %6, %6, %8\nv_add_u32 %7, %7, %8\n"
KERNEL(u_cndmask_mixed_vcc, unsigned, M4("vcc") M4("vcc") M4("vcc") M4("vcc") M4("vcc") M4("vcc") M4("vcc") M4("vcc"), "memory")
KERNEL(u_cndmask_mixed_sgpr, unsigned, "s_mov_b64 s[20:21], 0x5555\n" M4("s[20:21]") M4("s[20:21]") M4("s[20:21]") M4("s[20:21]") M4("s[20:21]") M4("s[20:21]") M4("s[20:21]") M4("s[20:21]"), "memory", "s20", "s21")
KERNEL(u_and_b32, unsigned, A64("v_and_b32", ", %8"), "memory")
KERNEL(u_or_b32, unsigned, A64("v_or_b32", ", %8"), "memory")
KERNEL(u_and_or_b32, unsigned, A64("v_and_or_b32", ", %8, %9"), "memory")
KERNEL(u_add_u32, unsigned, A64("v_add_u32", ", %8"), "memory")
KERNEL(u_lshlrev_b32, unsigned, A64("v_lshlrev_b32", ", 1"), "memory")
KERNEL(u_lshl_add_u32, unsigned, A64("v_lshl_add_u32", ", 1, %8"), "memory")
KERNEL(u_min_u32, unsigned, A64("v_min_u32", ", %8"), "memory")
KERNEL(u_max_i32, unsigned, A64("v_max_i32", ", %8"), "memory")
KERNEL(u_bfe_u32, unsigned, A64("v_bfe_u32", ", 3, 8"), "memory")
KERNEL(u_perm_b32, unsigned, A64("v_perm_b32", ", %8, %9"), "memory")
KERNEL(u_mov_b32, unsigned, "v_mov_b32 %0, %8\nv_mov_b32 %1, %9\nv_mov_b32 %2, %8\nv_mov_b32 %3, %9\nv_mov_b32 %4, %8\nv_mov_b32 %5, %9\nv_mov_b32 %6, %8\nv_mov_b32 %7, %9\n" A8("v_mov_b32", "") A8("v_mov_b32", "") A8("v_mov_b32", "") A8("v_mov_b32", "") A8("v_mov_b32", "") A8("v_mov_b32", "") A8("v_mov_b32", ""), "memory")
KERNEL(u_cvt_ubyte1, float, A64("v_cvt_f32_ubyte1", ""), "memory")
KERNEL(u_sad_u32, unsigned, A64("v_sad_u32", ", %8, %9"), "memory")
KERNEL(u_mad_u32_u24, unsigned, A64("v_mad_u32_u24", ", %8, %9"), "memory")
KERNEL(u_dot2c, float, A64("v_dot2c_f32_f16", ", %8"), "memory")
// SALU next to VALU: do scalar instructions share the issue slot?  64 v_fma_f32 interleaved with 64 s_add_u32
KERNEL(u_fma_plus_salu, float, A8("v_fma_f32", ", %8, %9\ns_add_u32 s20, s20, 1") A8("v_fma_f32", ", %8, %9\ns_add_u32 s21, s21, 1") A8("v_fma_f32", ", %8, %9\ns_add_u32 s20, s20, 1") A8("v_fma_f32", ", %8, %9\ns_add_u32 s21, s21, 1") A8("v_fma_f32", ", %8, %9\ns_add_u32 s20, s20, 1") A8("v_fma_f32", ", %8, %9\ns_add_u32 s21, s21, 1") A8("v_fma_f32", ", %8, %9\ns_add_u32 s20, s20, 1") A8("v_fma_f32", ", %8, %9\ns_add_u32 s21, s21, 1"), "memory", "s20", "s21", "scc")
KERNEL(u_fma_only, float, A64("v_fma_f32", ", %8, %9"), "memory")

int main() {
  float* d;
  (void)hipMalloc(&d, sizeof(float) * 256 * 8 * 256);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
  struct K { const char* name; void (*fn)(float*); };
  const K ks[] = {{"v_fma_f32", u_fma_only}, {"v_fma_f32 -|x|", u_fma_abs}, {"v_fmac_f32", u_fmac}, {"v_add_f32", u_add_f32}, {"v_sub_f32", u_sub_f32}, {"v_mul_f32", u_mul_f32},
                  {"v_max_f32", u_max_f32}, {"v_max3_f32", u_max3_f32}, {"v_min3_f32", u_min3_f32}, {"v_cmp_le_f32", u_cmp_le_f32}, {"v_cndmask_b32", u_cndmask},
                  {"v_cndmask_b32 (mask in an SGPR pair)", u_cndmask_sgpr}, {"v_cndmask_b32 (vcc written by v_cmp)", u_cndmask_init}, {"v_cndmask_b32_e64 ... vcc", u_cndmask_e64vcc}, {"16 v_cndmask(vcc) + 48 v_add_u32", u_cndmask_mixed_vcc}, {"16 v_cndmask(SGPR pair) + 48 v_add_u32", u_cndmask_mixed_sgpr}, {"v_and_b32", u_and_b32}, {"v_or_b32", u_or_b32}, {"v_and_or_b32", u_and_or_b32}, {"v_add_u32", u_add_u32}, {"v_lshlrev_b32", u_lshlrev_b32},
                  {"v_lshl_add_u32", u_lshl_add_u32}, {"v_min_u32", u_min_u32}, {"v_max_i32", u_max_i32}, {"v_bfe_u32", u_bfe_u32}, {"v_perm_b32", u_perm_b32},
                  {"v_mov_b32", u_mov_b32}, {"v_cvt_f32_ubyte1", u_cvt_ubyte1}, {"v_sad_u32", u_sad_u32}, {"v_mad_u32_u24", u_mad_u32_u24}, {"v_dot2c_f32_f16", u_dot2c},
                  {"v_fma_f32 + s_add_u32 (per pair)", u_fma_plus_salu}};
  for (const K& k : ks) {
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(k.fn, dim3(256 * 8), dim3(256), 0, 0, d);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      (void)hipEventElapsedTime(&ms, e0, e1);
    }
    // per SIMD: 8 waves x ITERS x 64 instructions
    const double inst = 8.0 * ITERS * 64.0;
    printf("%-34s %8.3f ms  %5.2f cycles per wave-instruction and SIMD at 2.4 GHz\n", k.name, ms, ms * 1e-3 * 2.4e9 / inst);
  }
  return 0;
}
