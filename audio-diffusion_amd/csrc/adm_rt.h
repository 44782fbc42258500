// adm_rt.h — thin runtime layer shared by every kernel file.
// Product build: hipcc --offload-arch=gfx950 (HIP runtime, real kernels).
// -DADM_EMU (tests only): the same sources compile with g++ against tests/emu/hip_emu.h.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <string>

#if defined(ADM_EMU)
#include "hip_emu.h"
#define ADM_LAUNCH(kern, grid, block, shmem, stream, ...) \
  adm_emu::launch((grid), (block), (shmem), [=]() { kern(__VA_ARGS__); })
#define ADM_DYN_SMEM(type, name) type* name = reinterpret_cast<type*>(adm_emu::S().dyn_smem)
#define ADM_UNROLL
// async global->LDS copy of 16 B per lane: LDS destination = wave-uniform base + lane*16 (guide §5)
#define ADM_GLDS16(gptr, lds_wave_base) \
  memcpy(reinterpret_cast<char*>(lds_wave_base) + (adm_emu::flat_tid() & 63) * 16, (gptr), 16)
// workgroup barrier that lets the newest N vector-memory loads of this wave stay in flight (emulation: plain barrier)
#define ADM_BARRIER_KEEP_VMEM(N) __syncthreads()
#define ADM_GLDS16_RAW(gptr, lds_wave_base) ADM_GLDS16(gptr, lds_wave_base)
#define ADM_WAIT_VMEM(N) ((void)0)
#define ADM_SCHED_FENCE() ((void)0)
#define ADM_RCP(x) (1.0f / (x))
#define ADM_UNIFORM(x) (x)
// bf16 operand plumbing (k_conv_bf16.hip): two floats -> one dword of 2 x bf16 (round-to-nearest-even, low half = a),
// v_mfma_f32_32x32x16_bf16 on 4-dword operands, v_alignbit_b32
#define ADM_PK_BF16(a, b) adm_emu::pk_bf16((a), (b))
#define ADM_MFMA_BF16(a, b, c) adm_emu::mfma_f32_32x32x16_op16<false>((a), (b), (c))
#define ADM_PK_F16(a, b) adm_emu::pk_f16((a), (b))
#define ADM_MFMA_F16(a, b, c) adm_emu::mfma_f32_32x32x16_op16<true>((a), (b), (c))
#define ADM_ALIGNBIT(hi, lo, sh) ((unsigned)((((uint64_t)(hi) << 32) | (uint64_t)(lo)) >> (sh)))
#define ADM_OPAQUE_V(x) ((void)0)
// round 4 (k_conv_bf16b.hip): LDS-DMA hidden from the compiler's wait-count insertion, source = wave-uniform base + per-lane
// byte offset; ds_read_b64_tr_b16 (each lane of a 16-lane group reads 8 bytes at its own address, the 16 x 4 halfwords are
// handed out transposed: lane l, element j <- lane (l >> 2) + 4 j, element l & 3)
#define ADM_LDS_ADDR(ptr) ((unsigned)(reinterpret_cast<const unsigned char*>(ptr) - adm_emu::S().dyn_smem))
#define ADM_GLDS16_ASM(gbase, off_bytes, lds_addr) \
  memcpy(adm_emu::S().dyn_smem + (lds_addr) + (adm_emu::flat_tid() & 63) * 16, reinterpret_cast<const char*>(gbase) + (off_bytes), 16)
#define ADM_DS_READ_TR16_B64(lds_ptr) adm_emu::ds_read_tr16_b64(lds_ptr)
#define ADM_BARRIER_LGKM() __syncthreads()
// LDS hand-over between the lanes of ONE wave (hardware: a wave's LDS operations execute in order, nothing to do; the
// emulator runs every lane as its own fiber and needs the rendezvous)
#define ADM_WAVE_LDS_ORDER() adm_emu::wave_sync()
// streaming (non-temporal) 16-byte accesses: plain ones here
template <class T> static inline T adm_ld_nt(const T* p) { return *p; }
template <class T> static inline void adm_st_nt(T* p, const T& v) { *p = v; }
#else
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 adm_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 adm_bf16x2 __attribute__((ext_vector_type(2)));
// two floats -> one dword of 2 x bf16 (v_cvt_pk_bf16_f32: round-to-nearest-even; low half = a)
#define ADM_PK_BF16(a, b) __builtin_bit_cast(unsigned, adm_bf16x2{(__bf16)(a), (__bf16)(b)})
// v_mfma_f32_32x32x16_bf16 on 4-dword operands: lane l holds A[i = l & 31][k = 8 (l >> 5) + 0..7] (B likewise with j);
// any k assignment is valid as long as A and B use the same one. C/D: col = l & 31, row = (r & 3) + 8 (r >> 2) + 4 (l >> 5).
#define ADM_MFMA_BF16(a, b, c) \
  __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(adm_bf16x8, (a)), __builtin_bit_cast(adm_bf16x8, (b)), (c), 0, 0, 0)
// the same two operations on IEEE binary16 operands (`--mixed_precision fp16`): v_cvt_f16_f32 (RNE) x2 + pack, v_mfma_f32_32x32x16_f16
typedef _Float16 adm_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 adm_f16x2 __attribute__((ext_vector_type(2)));
#define ADM_PK_F16(a, b) __builtin_bit_cast(unsigned, adm_f16x2{(_Float16)(a), (_Float16)(b)})
#define ADM_MFMA_F16(a, b, c) \
  __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(adm_f16x8, (a)), __builtin_bit_cast(adm_f16x8, (b)), (c), 0, 0, 0)
#define ADM_ALIGNBIT(hi, lo, sh) __builtin_amdgcn_alignbit((hi), (lo), (sh))
// makes a per-lane value opaque to the optimiser (no instruction): stops it from folding a loop-invariant lane offset into
// dozens of pre-computed 64-bit addresses that then live in registers across the whole loop
#define ADM_OPAQUE_V(x) asm volatile("" : "+v"(x))
// round 4 (k_conv_bf16b.hip): LDS-DMA of 16 B per lane written as inline asm — invisible to hipcc's wait-count insertion, so
// the kernel counts the vector-memory queue by hand (ADM_WAIT_VMEM) and nothing is drained behind its back.  Source = wave-
// uniform 64-bit base (SGPR pair) + per-lane 32-bit byte offset; destination = wave-uniform LDS BYTE ADDRESS (ADM_LDS_ADDR of a dynamic-LDS
// pointer, plus uniform integer offsets: no generic -> LDS pointer cast per instruction) + lane * 16.  M0 is
// compiler-reserved: saved and restored inside the statement (cdna_hip_programming.md §5.7).
#define ADM_LDS_ADDR(ptr) ((unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(ptr))
// (the base is passed through readfirstlane: a value the compiler happens to hold in VGPRs — e.g. a subexpression it shares
// with per-lane address arithmetic — would otherwise be substituted for the "s" operand as it is; the two s_mov + s_nop 2 in
// front of the load are also the five wait states a VALU-written SGPR needs before a VMEM instruction reads it)
__device__ __forceinline__ const void* adm_uniform_ptr(const void* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (const void*)(((unsigned long long)hi << 32) | lo);
}
#define ADM_GLDS16_ASM(gbase, off_bytes, lds_addr)                                                                   \
  do {                                                                                                               \
    unsigned keep_;                                                                                                  \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 2\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" \
                 : "=&s"(keep_) : "v"((unsigned)(off_bytes)), "s"(adm_uniform_ptr(gbase)),                           \
                   "s"(__builtin_amdgcn_readfirstlane((unsigned)(lds_addr))) : "memory");                            \
  } while (0)
// ds_read_b64_tr_b16: 8 bytes per lane at the lane's own (8-byte aligned) LDS address, transposed inside 16-lane groups
typedef short adm_s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned adm_u32x2 __attribute__((ext_vector_type(2)));
#define ADM_DS_READ_TR16_B64(lds_ptr) \
  __builtin_bit_cast(adm_u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) adm_s16x4*)(lds_ptr)))
// raw workgroup barrier that drains this wave's LDS traffic only (vector memory keeps flying)
#define ADM_BARRIER_LGKM() ADM_BARRIER_KEEP_VMEM(63)
// LDS hand-over between the lanes of ONE wave: LDS executes a wave's operations in order; only the compiler must not reorder
#define ADM_WAVE_LDS_ORDER() __builtin_amdgcn_wave_barrier()
// Streaming (non-temporal) 16-byte global accesses for passes over tensors far larger than the L2 (4 MB per XCD): the lines are
// not kept, which is worth +10 % (loads) / +16 % (loads and stores) on a 1.6 GB GroupNorm-backward pass (tools/microbench/
// gnbwd_bench.hip: 5.6 -> 6.6 TB/s). T = float4 / u32x4.
template <class T> __device__ __forceinline__ T adm_ld_nt(const T* p) {
  typedef unsigned adm_nt4 __attribute__((ext_vector_type(4)));
  static_assert(sizeof(T) == 16, "adm_ld_nt: 16-byte types");
  const adm_nt4 v = __builtin_nontemporal_load(reinterpret_cast<const adm_nt4*>(p));
  return __builtin_bit_cast(T, v);
}
template <class T> __device__ __forceinline__ void adm_st_nt(T* p, const T& v) {
  typedef unsigned adm_nt4 __attribute__((ext_vector_type(4)));
  static_assert(sizeof(T) == 16, "adm_st_nt: 16-byte types");
  __builtin_nontemporal_store(__builtin_bit_cast(adm_nt4, v), reinterpret_cast<adm_nt4*>(p));
}
#define ADM_LAUNCH(kern, grid, block, shmem, stream, ...) \
  hipLaunchKernelGGL(kern, (grid), (block), (shmem), (stream), __VA_ARGS__)
#define ADM_DYN_SMEM(type, name)                                              \
  extern __shared__ __attribute__((aligned(16))) unsigned char adm_dyn_smem_[]; \
  type* name = reinterpret_cast<type*>(adm_dyn_smem_)
#define ADM_UNROLL _Pragma("unroll")
#define ADM_GLDS16(gptr, lds_wave_base)                                                               \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),             \
                                   (__attribute__((address_space(3))) void*)(lds_wave_base), 16, 0, 0)
// The same LDS-DMA written as inline asm: invisible to hipcc's wait-count insertion. Next to a builtin LDS-DMA the compiler
// waits vmcnt(0) at the first use of ANY plain load (guide §5, "mixing load kinds"); hidden like this, its counted waits
// for the wave's plain loads only see those loads — they over-wait by the DMA pieces in between (safe: vector memory
// returns in order) and the DMA itself is drained by hand with ADM_WAIT_VMEM / ADM_BARRIER_KEEP_VMEM.
#define ADM_GLDS16_RAW(gptr, lds_wave_base)                                                                        \
  do {                                                                                                             \
    const unsigned m0v_ = __builtin_amdgcn_readfirstlane(                                                          \
        (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(lds_wave_base));                            \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(m0v_), "v"(gptr) : "memory"); \
  } while (0)
#define ADM_WAIT_VMEM(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
// Raw s_barrier with a COUNTED vmcnt: everything older than the newest N VMEM operations of this wave (in particular an
// LDS-DMA issued before them) has landed, the newest N loads keep flying across the barrier; LDS traffic is drained.
// (__syncthreads() would emit vmcnt(0) whenever an LDS-DMA is outstanding and drain the prefetch as well; guide §5.)
#define ADM_SCHED_FENCE()                  \
  do {                                     \
    asm volatile("" ::: "memory");         \
    __builtin_amdgcn_sched_barrier(0);     \
  } while (0)
// v_rcp_f32 (1 ulp) instead of the ~12-instruction IEEE division sequence that `/` and __fdividef expand to
#if defined(ADM_PRECISE_MATH)   // measurement build (tools/accuracy_probe.py): IEEE division and libm's expf where the product uses v_rcp_f32 / v_exp_f32
#define ADM_RCP(x) (1.0f / (x))
#define __expf(x) expf(x)
#else
#define ADM_RCP(x) __builtin_amdgcn_rcpf(x)
#endif
// a value known to be wave-uniform, moved to an SGPR so that tests on it become scalar branches
#define ADM_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
#define ADM_BARRIER_KEEP_VMEM(N)                                            \
  do {                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                      \
    asm volatile("s_waitcnt vmcnt(" #N ") lgkmcnt(0)" ::: "memory");        \
    __builtin_amdgcn_s_barrier();                                           \
    asm volatile("" ::: "memory");                                          \
    __builtin_amdgcn_sched_barrier(0);                                      \
  } while (0)
#endif

// 16-bit MFMA operand format as a compile-time flag of the kernels (F16_ false: bf16, true: IEEE binary16)
#define ADM_PK16(F16_, a, b) ((F16_) ? ADM_PK_F16((a), (b)) : ADM_PK_BF16((a), (b)))
#define ADM_MFMA16(F16_, a, b, c) ((F16_) ? ADM_MFMA_F16((a), (b), (c)) : ADM_MFMA_BF16((a), (b), (c)))

namespace adm {

// ---- error plumbing: C-ABI functions return int (0 ok) and record a message ------------------
void set_error(const std::string& msg);
const char* last_error();

#define ADM_FAIL(msg)                                                          \
  do {                                                                         \
    ::adm::set_error(std::string(__FILE__) + ":" + std::to_string(__LINE__) + ": " + (msg)); \
    return -1;                                                                 \
  } while (0)
#define ADM_REQUIRE(cond, msg) \
  do {                         \
    if (!(cond)) ADM_FAIL(msg); \
  } while (0)

#if defined(ADM_EMU)
#define ADM_HIP_OK(expr) (void)(expr)
#define ADM_CHECK_LAUNCH() 0
inline int dmalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? 0 : -1; }
inline void dfree(void* p) { free(p); }
inline int copy_h2d(void* d, const void* s, size_t n, hipStream_t) { memcpy(d, s, n); return 0; }
inline int copy_d2h(void* d, const void* s, size_t n, hipStream_t) { memcpy(d, s, n); return 0; }
inline int copy_d2d(void* d, const void* s, size_t n, hipStream_t) { memmove(d, s, n); return 0; }
inline int dmemset(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return 0; }
inline int stream_sync(hipStream_t) { return 0; }
#else
#define ADM_HIP_OK(expr)                                                              \
  do {                                                                                \
    hipError_t e_ = (expr);                                                           \
    if (e_ != hipSuccess) ADM_FAIL(std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)
inline int check_launch_() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error(std::string("kernel launch: ") + hipGetErrorString(e)); return -1; }
  return 0;
}
#define ADM_CHECK_LAUNCH() ::adm::check_launch_()
inline int dmalloc(void** p, size_t n) { ADM_HIP_OK(hipMalloc(p, n ? n : 1)); return 0; }
inline void dfree(void* p) { (void)hipFree(p); }
inline int copy_h2d(void* d, const void* s, size_t n, hipStream_t st) { ADM_HIP_OK(hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, st)); return 0; }
inline int copy_d2h(void* d, const void* s, size_t n, hipStream_t st) { ADM_HIP_OK(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, st)); return 0; }
inline int copy_d2d(void* d, const void* s, size_t n, hipStream_t st) { ADM_HIP_OK(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToDevice, st)); return 0; }
inline int dmemset(void* d, int v, size_t n, hipStream_t st) { ADM_HIP_OK(hipMemsetAsync(d, v, n, st)); return 0; }
inline int stream_sync(hipStream_t st) { ADM_HIP_OK(hipStreamSynchronize(st)); return 0; }
#endif

#define ADM_TRY(expr)        \
  do {                       \
    int rc_ = (expr);        \
    if (rc_ != 0) return rc_; \
  } while (0)

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

}  // namespace adm
