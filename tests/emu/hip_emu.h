// hip_emu.h — TEST INFRASTRUCTURE ONLY. A lock-step fiber emulator of the HIP execution model
// (workgroups, 64-lane waves, __syncthreads, wave shuffles, f32 MFMA fragment semantics) so that
// the *same* kernel sources under audio-diffusion_amd/csrc compile with g++ (-DADM_EMU) and can be
// checked against the oracle inside this GPU-less container. The product never builds or loads
// this; the shipped library is compiled by hipcc for gfx950 only (see __graft_entry__.build()).
//
// Model: each workgroup's threads are ucontext fibers on one OS thread, scheduled round-robin and
// switched only at collectives, so a missing barrier shows up deterministically (thread 0 runs
// ahead). Workgroups are distributed over OS threads; `__shared__` becomes `static thread_local`.
// MFMA lane->element maps follow /opt/skills/guides/cdna_hip_programming.md §3.
#pragma once

#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(8) float2 { float x, y; };
struct alignas(16) double2 { double x, y; };
struct alignas(16) int4 { int x, y, z, w; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline double2 make_double2(double x, double y) { return double2{x, y}; }

typedef int hipError_t;
typedef void* hipStream_t;
#define hipSuccess 0

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static thread_local
#define __launch_bounds__(...)
#define __restrict__ __restrict

namespace adm_emu {

// Minimal x86-64 SysV context switch (callee-saved registers + stack pointer); ~10 ns instead of the
// sigprocmask syscall inside glibc's swapcontext. Weak so every translation unit may emit it.
extern "C" void adm_emu_ctx_switch(void** save_sp, void* load_sp);
__asm__(R"(
.text
.weak adm_emu_ctx_switch
.type adm_emu_ctx_switch,@function
adm_emu_ctx_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size adm_emu_ctx_switch,.-adm_emu_ctx_switch
)");

struct Fiber {
  void* sp = nullptr;
  char* stack = nullptr;
  bool done = true;
  dim3 tid;
};
struct Wave {
  int nlive = 0, arrived = 0;
  unsigned gen = 0;
  uint64_t xch[64];
  float a[64], b[64];
  uint32_t a8[64][4], b8[64][4];   // bf16 MFMA operands (8 x bf16 per lane)
};
struct State {
  // per worker thread; launch() creates its workers anew, so the fiber stacks must die with the thread
  // (they used to leak: ~256 MiB per launch of a 512-thread kernel, tens of GiB over a test session)
  ~State() {
    for (Fiber& f : fibers) free(f.stack);
  }
  dim3 grid, block, bid;
  std::vector<Fiber> fibers;
  std::vector<Wave> waves;
  int cur = 0, live = 0, bar_arrived = 0;
  unsigned bar_gen = 0;
  void* sched_sp = nullptr;
  unsigned char* dyn_smem = nullptr;
  void (*entry)(void*) = nullptr;
  void* entry_arg = nullptr;
};
inline State& S() {
  static thread_local State s;
  return s;
}
static constexpr size_t kStack = 96 * 1024;

inline void yield() {
  State& s = S();
  adm_emu_ctx_switch(&s.fibers[s.cur].sp, s.sched_sp);
}
inline int flat_tid() {
  State& s = S();
  const dim3& t = s.fibers[s.cur].tid;
  return t.x + s.block.x * (t.y + s.block.y * t.z);
}
inline void block_sync() {
  State& s = S();
  unsigned g = s.bar_gen;
  if (++s.bar_arrived >= s.live) {
    s.bar_arrived = 0;
    s.bar_gen++;
  } else {
    while (s.bar_gen == g) yield();
  }
}
inline void wave_sync() {
  State& s = S();
  Wave& w = s.waves[flat_tid() >> 6];
  unsigned g = w.gen;
  if (++w.arrived >= w.nlive) {
    w.arrived = 0;
    w.gen++;
  } else {
    unsigned spins = 0;
    while (w.gen == g) {
      yield();
      if (++spins > 1000000u) {
        fprintf(stderr, "adm_emu: divergent wave collective (deadlock) in block (%u,%u,%u)\n", s.bid.x, s.bid.y, s.bid.z);
        abort();
      }
    }
  }
}
inline void fiber_main() {
  State& s = S();
  s.entry(s.entry_arg);
  Fiber& f = s.fibers[s.cur];
  f.done = true;
  s.live--;
  Wave& w = s.waves[flat_tid() >> 6];
  w.nlive--;
  if (w.nlive > 0 && w.arrived >= w.nlive) { w.arrived = 0; w.gen++; }
  if (s.live > 0 && s.bar_arrived >= s.live) { s.bar_arrived = 0; s.bar_gen++; }
  adm_emu_ctx_switch(&f.sp, s.sched_sp);
  abort();  // a finished fiber is never resumed
}
inline void run_block(dim3 grid, dim3 block, dim3 bid, size_t shmem, void (*entry)(void*), void* arg) {
  State& s = S();
  s.grid = grid; s.block = block; s.bid = bid;
  s.entry = entry; s.entry_arg = arg;
  int n = block.x * block.y * block.z;
  if ((int)s.fibers.size() < n) s.fibers.resize(n);
  s.waves.assign((n + 63) / 64, Wave());
  std::vector<unsigned char> smem(shmem + 64);
  s.dyn_smem = (unsigned char*)(((uintptr_t)smem.data() + 63) & ~(uintptr_t)63);
  s.live = n; s.bar_arrived = 0; s.bar_gen = 0;
  for (int i = 0; i < n; ++i) {
    Fiber& f = s.fibers[i];
    if (!f.stack) f.stack = (char*)malloc(kStack);
    f.done = false;
    f.tid = dim3(i % block.x, (i / block.x) % block.y, i / (block.x * block.y));
    s.waves[i >> 6].nlive++;
    // initial frame: [r15 r14 r13 r12 rbx rbp] [ret -> fiber_main] [fake return address]; after the `ret`
    // rsp % 16 == 8, exactly as at a normal function entry.
    void** top = (void**)(((uintptr_t)f.stack + kStack) & ~(uintptr_t)15);
    *--top = nullptr;
    *--top = (void*)(void (*)())fiber_main;
    for (int r = 0; r < 6; ++r) *--top = nullptr;
    f.sp = (void*)top;
  }
  while (s.live > 0) {
    for (int i = 0; i < n; ++i) {
      if (s.fibers[i].done) continue;
      s.cur = i;
      adm_emu_ctx_switch(&s.sched_sp, s.fibers[i].sp);
    }
  }
}
template <class F>
void launch(dim3 grid, dim3 block, size_t shmem, F body) {
  struct Tramp { static void call(void* p) { (*(F*)p)(); } };
  size_t nblk = (size_t)grid.x * grid.y * grid.z;
  static int nthr_env = getenv("ADM_EMU_THREADS") ? atoi(getenv("ADM_EMU_THREADS")) : 0;
  unsigned nthr = nthr_env > 0 ? nthr_env : std::thread::hardware_concurrency();
  if (nthr > nblk) nthr = (unsigned)nblk;
  if (nthr < 1) nthr = 1;
  std::atomic<size_t> next{0};
  auto worker = [&]() {
    for (;;) {
      size_t b = next.fetch_add(1);
      if (b >= nblk) break;
      dim3 bid(b % grid.x, (b / grid.x) % grid.y, b / ((size_t)grid.x * grid.y));
      F local = body;
      run_block(grid, block, bid, shmem, &Tramp::call, &local);
    }
  };
  if (nthr == 1) { worker(); return; }
  std::vector<std::thread> th;
  for (unsigned i = 0; i < nthr; ++i) th.emplace_back(worker);
  for (auto& t : th) t.join();
}

template <class T>
inline T shfl_idx(T v, int src) {
  static_assert(sizeof(T) <= 8, "shfl payload");
  State& s = S();
  int lane = flat_tid() & 63;
  Wave& w = s.waves[flat_tid() >> 6];
  uint64_t raw = 0;
  memcpy(&raw, &v, sizeof(T));
  w.xch[lane] = raw;
  wave_sync();
  T r;
  memcpy(&r, &w.xch[src & 63], sizeof(T));
  wave_sync();
  return r;
}
}  // namespace adm_emu

#define threadIdx (adm_emu::S().fibers[adm_emu::S().cur].tid)
#define blockIdx (adm_emu::S().bid)
#define blockDim (adm_emu::S().block)
#define gridDim (adm_emu::S().grid)

static inline void __syncthreads() { adm_emu::block_sync(); }
template <class T> static inline T __shfl_xor(T v, int m, int = 64) { return adm_emu::shfl_idx(v, (adm_emu::flat_tid() & 63) ^ m); }
template <class T> static inline T __shfl_down(T v, int d, int = 64) {
  int l = adm_emu::flat_tid() & 63;
  return adm_emu::shfl_idx(v, l + d < 64 ? l + d : l);
}
template <class T> static inline T __shfl(T v, int src, int = 64) { return adm_emu::shfl_idx(v, src); }

typedef float f32x16 __attribute__((vector_size(64)));
typedef float f32x4 __attribute__((vector_size(16)));

// v_mfma_f32_32x32x2_f32: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]; D col=l&31,row=(r&3)+8*(r>>2)+4*(l>>5);
// k-ordered fmaf chain (bitwise-equal to the hardware per the guide §3).
static inline f32x16 __builtin_amdgcn_mfma_f32_32x32x2f32(float a, float b, f32x16 c, int, int, int) {
  using namespace adm_emu;
  int lane = flat_tid() & 63;
  Wave& w = S().waves[flat_tid() >> 6];
  w.a[lane] = a; w.b[lane] = b;
  wave_sync();
  f32x16 d;
  int j = lane & 31;
  for (int r = 0; r < 16; ++r) {
    int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    d[r] = fmaf(w.a[i + 32], w.b[j + 32], fmaf(w.a[i], w.b[j], c[r]));
  }
  wave_sync();
  return d;
}
// v_mfma_f32_16x16x4_f32: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15]; D col=l&15,row=4*(l>>4)+r.
static inline f32x4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, f32x4 c, int, int, int) {
  using namespace adm_emu;
  int lane = flat_tid() & 63;
  Wave& w = S().waves[flat_tid() >> 6];
  w.a[lane] = a; w.b[lane] = b;
  wave_sync();
  f32x4 d;
  int j = lane & 15;
  for (int r = 0; r < 4; ++r) {
    int i = 4 * (lane >> 4) + r;
    float acc = c[r];
    for (int k = 0; k < 4; ++k) acc = fmaf(w.a[i + 16 * k], w.b[j + 16 * k], acc);
    d[r] = acc;
  }
  wave_sync();
  return d;
}

typedef unsigned u32x4 __attribute__((vector_size(16)));
namespace adm_emu {
static inline uint32_t bf16_bits(float f) {        // round-to-nearest-even, NaN kept quiet (v_cvt_pk_bf16_f32)
  uint32_t u; memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
static inline unsigned pk_bf16(float a, float b) { return bf16_bits(a) | (bf16_bits(b) << 16); }
static inline float bf16_val(uint32_t dword, int hi) { uint32_t u = hi ? (dword & 0xffff0000u) : (dword << 16); float f; memcpy(&f, &u, 4); return f; }
// IEEE binary16 <-> binary32 (round-to-nearest-even on the way down, as v_cvt_f16_f32; subnormals kept)
static inline uint32_t f16_bits(float f) {
  uint32_t u; memcpy(&u, &f, 4);
  const uint32_t sign = (u >> 16) & 0x8000u, ax = u & 0x7fffffffu;
  if (ax > 0x7f800000u) return sign | 0x7e00u;                       // NaN
  if (ax >= 0x477ff000u) return sign | 0x7c00u;                      // >= 65520 rounds to inf (also inf)
  if (ax < 0x33000001u) return sign;                                 // < 2^-25 (or exactly 2^-25: ties to even = 0)
  int e = (int)(ax >> 23) - 127;
  uint32_t m = (ax & 0x7fffffu) | 0x800000u;                         // 24-bit significand
  int shift = e < -14 ? 13 + (-14 - e) : 13;                         // bits dropped (subnormal: more)
  uint32_t half = m >> shift, rem = m & ((1u << shift) - 1u), mid = 1u << (shift - 1);
  if (rem > mid || (rem == mid && (half & 1u))) ++half;
  if (e < -14) return sign | half;                                   // subnormal (a carry into 0x400 is the smallest normal)
  return sign | (uint32_t)(((e + 15) << 10) + (half - 0x400u));      // a carry out of the significand bumps the exponent
}
static inline float f16_val(uint32_t dword, int hi) {
  const uint32_t h = hi ? (dword >> 16) : (dword & 0xffffu);
  const uint32_t sign = (h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 0x3ffu;
  uint32_t u;
  if (e == 0) {
    if (m == 0) u = sign;
    else { int k = 0; uint32_t mm = m; while (!(mm & 0x400u)) { mm <<= 1; ++k; } u = sign | (uint32_t)((113 - k) << 23) | ((mm & 0x3ffu) << 13); }
  } else if (e == 31) u = sign | 0x7f800000u | (m << 13);
  else u = sign | ((e + 112u) << 23) | (m << 13);
  float f; memcpy(&f, &u, 4); return f;
}
static inline unsigned pk_f16(float a, float b) { return f16_bits(a) | (f16_bits(b) << 16); }
// v_mfma_f32_32x32x16_bf16 / _f16: A[i=l&31][k=8(l>>5)+e], B[k][j=l&31]; exact products, fp32 accumulation in k order
template <bool F16>
static inline f32x16 mfma_f32_32x32x16_op16(u32x4 a, u32x4 b, f32x16 c) {
  int lane = flat_tid() & 63;
  Wave& w = S().waves[flat_tid() >> 6];
  for (int q = 0; q < 4; ++q) { w.a8[lane][q] = a[q]; w.b8[lane][q] = b[q]; }
  wave_sync();
  f32x16 d;
  int j = lane & 31;
  for (int r = 0; r < 16; ++r) {
    int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    float acc = c[r];
    for (int h = 0; h < 2; ++h)
      for (int e = 0; e < 8; ++e) {
        const float x = F16 ? f16_val(w.a8[i + 32 * h][e >> 1], e & 1) : bf16_val(w.a8[i + 32 * h][e >> 1], e & 1);
        const float y = F16 ? f16_val(w.b8[j + 32 * h][e >> 1], e & 1) : bf16_val(w.b8[j + 32 * h][e >> 1], e & 1);
        acc = fmaf(x, y, acc);
      }
    d[r] = acc;
  }
  wave_sync();
  return d;
}
}  // namespace adm_emu

// ds_read_b64_tr_b16 (gfx950): every lane reads 8 bytes at its own LDS address; inside each 16-lane group the 16 x 4 halfwords
// are handed out transposed: lane l (0..15), element j (0..3) receives element l & 3 of lane (l >> 2) + 4 j.  With the 16 lanes
// on 128 consecutive bytes this is "column l of a row-major 4 x 16 matrix" (cdna_hip_programming.md §2).
typedef unsigned adm_u32x2 __attribute__((vector_size(8)));
namespace adm_emu {
static inline adm_u32x2 ds_read_tr16_b64(const void* p) {
  int lane = flat_tid() & 63;
  Wave& w = S().waves[flat_tid() >> 6];
  uint64_t raw;
  memcpy(&raw, p, 8);
  w.xch[lane] = raw;
  wave_sync();
  const int g = lane & ~15, l = lane & 15;
  uint16_t e[4];
  for (int j = 0; j < 4; ++j) e[j] = (uint16_t)(w.xch[g + (l >> 2) + 4 * j] >> (16 * (l & 3)));
  wave_sync();
  adm_u32x2 r;
  r[0] = (unsigned)e[0] | ((unsigned)e[1] << 16);
  r[1] = (unsigned)e[2] | ((unsigned)e[3] << 16);
  return r;
}
}  // namespace adm_emu

static inline float adm_emu_expf(float x) { return expf(x); }
#define __expf adm_emu_expf
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline double rsqrt(double x) { return 1.0 / sqrt(x); }
static inline float __fdividef(float a, float b) { return a / b; }
static inline float __fdiv_rn(float a, float b) { return a / b; }
static inline float __fmul_rn(float a, float b) { volatile float r = a * b; return r; }   // volatile: no fma contraction
static inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }
static inline void sincospi(double x, double* s, double* c) { *s = sin(M_PI * x); *c = cos(M_PI * x); }

template <class T> static inline T atomicAdd(T* p, T v) {
  if constexpr (std::is_integral<T>::value) {
    return __atomic_fetch_add(p, v, __ATOMIC_RELAXED);
  } else {
    using U = typename std::conditional<sizeof(T) == 4, uint32_t, uint64_t>::type;
    U* up = (U*)p;
    U old = __atomic_load_n(up, __ATOMIC_RELAXED), nw;
    T o;
    do {
      memcpy(&o, &old, sizeof(T));
      T n = o + v;
      memcpy(&nw, &n, sizeof(T));
    } while (!__atomic_compare_exchange_n(up, &old, nw, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
    return o;
  }
}
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline long long __double_as_longlong(double d) { long long u; memcpy(&u, &d, 8); return u; }
static inline double __longlong_as_double(long long u) { double d; memcpy(&d, &u, 8); return d; }
static inline unsigned long long atomicMax(unsigned long long* p, unsigned long long v) {
  unsigned long long old = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return old;
}
static inline unsigned atomicMax(unsigned* p, unsigned v) {
  unsigned old = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return old;
}
