// micro-test: ordered per-vertex accumulation by DPP row_shl ripple vs the LDS slab of k_persistent_pv
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#include <cstring>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// lanes: vertex segments inside 16-lane rows; first[lane], deg of the lane's vertex (degx = 255 for non-heads)
template <int J>
__device__ __forceinline__ void ripple_step_m(float& X, float& W1, float& W2, const float cx, const float a1, const float a2, const float b1,
                                              const float b2, const unsigned long long m) {
  asm volatile("s_mov_b64 exec, %[m]\n\t"
               "v_add_f32_dpp %[X], %[cx], %[X] row_shl:%[sh] row_mask:0xf bank_mask:0xf\n\t"
               "v_add_f32_dpp %[W1], %[a1], %[W1] row_shl:%[sh] row_mask:0xf bank_mask:0xf\n\t"
               "v_add_f32_dpp %[W2], %[a2], %[W2] row_shl:%[sh] row_mask:0xf bank_mask:0xf\n\t"
               "v_add_f32_dpp %[W1], %[b1], %[W1] row_shl:%[sh] row_mask:0xf bank_mask:0xf\n\t"
               "v_add_f32_dpp %[W2], %[b2], %[W2] row_shl:%[sh] row_mask:0xf bank_mask:0xf\n\t"
               : [X] "+v"(X), [W1] "+v"(W1), [W2] "+v"(W2)
               : [cx] "v"(cx), [a1] "v"(a1), [a2] "v"(a2), [b1] "v"(b1), [b2] "v"(b2), [m] "s"(m), [sh] "n"(J)
               : "exec");
}
template <int J>
__device__ __forceinline__ void ripple_step_n(float& X, float& W1, float& W2, const float cx, const float a1, const float a2, const float b1,
                                              const float b2) {
  asm volatile("v_add_f32_dpp %[X], %[cx], %[X] row_shl:%[sh] row_mask:0xf bank_mask:0xf\n\t"
               "v_add_f32_dpp %[W1], %[a1], %[W1] row_shl:%[sh] row_mask:0xf bank_mask:0xf\n\t"
               "v_add_f32_dpp %[W2], %[a2], %[W2] row_shl:%[sh] row_mask:0xf bank_mask:0xf\n\t"
               "v_add_f32_dpp %[W1], %[b1], %[W1] row_shl:%[sh] row_mask:0xf bank_mask:0xf\n\t"
               "v_add_f32_dpp %[W2], %[b2], %[W2] row_shl:%[sh] row_mask:0xf bank_mask:0xf\n\t"
               : [X] "+v"(X), [W1] "+v"(W1), [W2] "+v"(W2)
               : [cx] "v"(cx), [a1] "v"(a1), [a2] "v"(a2), [b1] "v"(b1), [b2] "v"(b2), [sh] "n"(J));
}
__device__ __forceinline__ void ripple_step_p(float& X, float& W1, float& W2, const float cx, const float a1, const float a2, const float b1,
                                              const float b2) {
  asm volatile("v_add_f32 %[X], %[cx], %[X]\n\t"
               "v_add_f32 %[W1], %[a1], %[W1]\n\t"
               "v_add_f32 %[W2], %[a2], %[W2]\n\t"
               "v_add_f32 %[W1], %[b1], %[W1]\n\t"
               "v_add_f32 %[W2], %[b2], %[W2]\n\t"
               : [X] "+v"(X), [W1] "+v"(W1), [W2] "+v"(W2)
               : [cx] "v"(cx), [a1] "v"(a1), [a2] "v"(a2), [b1] "v"(b1), [b2] "v"(b2));
}

template <int MODE>
__global__ void __launch_bounds__(64) k_ripple8(const int* first, const int* deg, const float* c5, float* out, long long* cyc, int reps) {
  const int lane = threadIdx.x;
  const int f = first[lane], d = deg[lane];
  const bool head = f == lane;
  const unsigned degx = head ? (unsigned)d : 255u;
  float cx = c5[lane * 5], a1 = c5[lane * 5 + 1], a2 = c5[lane * 5 + 2], b1 = c5[lane * 5 + 3], b2 = c5[lane * 5 + 4];
  float x = 1.0f + lane, w1 = 0.5f, w2 = -0.25f;
  float X = 0, W1 = 0, W2 = 0;
  unsigned long long m[8];
  for (int j = 1; j < 8; ++j) m[j] = __ballot(degx > (unsigned)j);
  const long long t0 = clock64();
  for (int r = 0; r < reps; ++r) {
    X = x + cx, W1 = (w1 + a1) + b1, W2 = (w2 + a2) + b2;
    asm volatile("" : "+v"(X), "+v"(W1), "+v"(W2));
    if (MODE == 1) {
      ripple_step_m<1>(X, W1, W2, cx, a1, a2, b1, b2, m[1]); ripple_step_m<2>(X, W1, W2, cx, a1, a2, b1, b2, m[2]);
      ripple_step_m<3>(X, W1, W2, cx, a1, a2, b1, b2, m[3]); ripple_step_m<4>(X, W1, W2, cx, a1, a2, b1, b2, m[4]);
      ripple_step_m<5>(X, W1, W2, cx, a1, a2, b1, b2, m[5]); ripple_step_m<6>(X, W1, W2, cx, a1, a2, b1, b2, m[6]);
      ripple_step_m<7>(X, W1, W2, cx, a1, a2, b1, b2, m[7]);
      asm volatile("s_mov_b64 exec, -1" ::: "exec");
    } else if (MODE == 2) {
      ripple_step_n<1>(X, W1, W2, cx, a1, a2, b1, b2); ripple_step_n<2>(X, W1, W2, cx, a1, a2, b1, b2);
      ripple_step_n<3>(X, W1, W2, cx, a1, a2, b1, b2); ripple_step_n<4>(X, W1, W2, cx, a1, a2, b1, b2);
      ripple_step_n<5>(X, W1, W2, cx, a1, a2, b1, b2); ripple_step_n<6>(X, W1, W2, cx, a1, a2, b1, b2);
      ripple_step_n<7>(X, W1, W2, cx, a1, a2, b1, b2);
    } else {
      for (int j = 1; j < 8; ++j) ripple_step_p(X, W1, W2, cx, a1, a2, b1, b2);
    }
    x = head ? X * 1e-30f + x : x;
  }
  const long long t1 = clock64();
  out[lane * 3] = X, out[lane * 3 + 1] = W1, out[lane * 3 + 2] = W2;
  if (lane == 0) cyc[blockIdx.x] = (t1 - t0) / reps;
}


#define PV_ADDS(J)                                                                                   \
  "v_add_f32_dpp %[X], %[cx], %[X] row_shl:" #J " row_mask:0xf bank_mask:0xf\n\t"                      \
  "v_add_f32_dpp %[W1], %[a1], %[W1] row_shl:" #J " row_mask:0xf bank_mask:0xf\n\t"                    \
  "v_add_f32_dpp %[W2], %[a2], %[W2] row_shl:" #J " row_mask:0xf bank_mask:0xf\n\t"                    \
  "v_add_f32_dpp %[W1], %[b1], %[W1] row_shl:" #J " row_mask:0xf bank_mask:0xf\n\t"                    \
  "v_add_f32_dpp %[W2], %[b2], %[W2] row_shl:" #J " row_mask:0xf bank_mask:0xf\n\t"
#define PV_RS4(J, J1, CUR, NXT) "s_cmp_le_u32 %[md], " #J "\n\ts_cbranch_scc1 9f\n\t" "v_cmp_lt_u32_e64 %[" #NXT "], " #J1 ", %[dx]\n\t" "s_mov_b64 exec, %[" #CUR "]\n\t" PV_ADDS(J)
#define PV_RS5(J, J1, CUR, NXT) "v_cmp_lt_u32_e64 %[" #NXT "], " #J1 ", %[dx]\n\t" "s_mov_b64 exec, %[" #CUR "]\n\t" PV_ADDS(J)
#define PV_RS6(J, M) "s_cmp_le_u32 %[md], " #J "\n\ts_cbranch_scc1 9f\n\t" "s_mov_b64 exec, %[" #M "]\n\t" PV_ADDS(J)
template <int MODE>
__global__ void __launch_bounds__(64) k_ripple_k(const int* first, const int* deg, const float* c5, float* out, long long* cyc, int reps, int maxdeg) {
  const int lane = threadIdx.x;
  const int f = first[lane], d = deg[lane];
  const bool head = f == lane;
  const unsigned degx = head ? (unsigned)d : 255u;
  float cx = c5[lane * 5], a1 = c5[lane * 5 + 1], a2 = c5[lane * 5 + 2], b1 = c5[lane * 5 + 3], b2 = c5[lane * 5 + 4];
  float x = 1.0f + lane, w1 = 0.5f, w2 = -0.25f;
  float X = 0, W1 = 0, W2 = 0;
  unsigned long long m1 = __ballot(degx > 1u), m2 = __ballot(degx > 2u), m3 = __ballot(degx > 3u), m4 = __ballot(degx > 4u), m5 = __ballot(degx > 5u),
                     m6 = __ballot(degx > 6u), m7 = __ballot(degx > 7u);
  const long long t0 = clock64();
  for (int r = 0; r < reps; ++r) {
    X = x + cx, W1 = (w1 + a1) + b1, W2 = (w2 + a2) + b2;
    unsigned long long ma, mb;
    if (MODE == 4) {
      asm volatile("s_nop 1\n\tv_cmp_lt_u32_e64 %[ma], 1, %[dx]\n\t" PV_RS4(1, 2, ma, mb) PV_RS4(2, 3, mb, ma) PV_RS4(3, 4, ma, mb) PV_RS4(4, 5, mb, ma)
                   PV_RS4(5, 6, ma, mb) PV_RS4(6, 7, mb, ma) PV_RS4(7, 8, ma, mb) "9:\n\ts_mov_b64 exec, -1"
                   : [X] "+v"(X), [W1] "+v"(W1), [W2] "+v"(W2), [ma] "=&s"(ma), [mb] "=&s"(mb)
                   : [cx] "v"(cx), [a1] "v"(a1), [a2] "v"(a2), [b1] "v"(b1), [b2] "v"(b2), [dx] "v"(degx), [md] "s"(maxdeg) : "scc");
    } else if (MODE == 5) {
      asm volatile("s_nop 1\n\tv_cmp_lt_u32_e64 %[ma], 1, %[dx]\n\t" PV_RS5(1, 2, ma, mb) PV_RS5(2, 3, mb, ma) PV_RS5(3, 4, ma, mb) PV_RS5(4, 5, mb, ma)
                   PV_RS5(5, 6, ma, mb) PV_RS5(6, 7, mb, ma) PV_RS5(7, 8, ma, mb) "s_mov_b64 exec, -1"
                   : [X] "+v"(X), [W1] "+v"(W1), [W2] "+v"(W2), [ma] "=&s"(ma), [mb] "=&s"(mb)
                   : [cx] "v"(cx), [a1] "v"(a1), [a2] "v"(a2), [b1] "v"(b1), [b2] "v"(b2), [dx] "v"(degx), [md] "s"(maxdeg) : "scc");
    } else {
      asm volatile("s_nop 1\n\t" PV_RS6(1, m1) PV_RS6(2, m2) PV_RS6(3, m3) PV_RS6(4, m4) PV_RS6(5, m5) PV_RS6(6, m6) PV_RS6(7, m7) "9:\n\ts_mov_b64 exec, -1"
                   : [X] "+v"(X), [W1] "+v"(W1), [W2] "+v"(W2)
                   : [cx] "v"(cx), [a1] "v"(a1), [a2] "v"(a2), [b1] "v"(b1), [b2] "v"(b2), [md] "s"(maxdeg), [m1] "s"(m1), [m2] "s"(m2), [m3] "s"(m3),
                     [m4] "s"(m4), [m5] "s"(m5), [m6] "s"(m6), [m7] "s"(m7) : "scc");
    }
    x = head ? X * 1e-30f + x : x;
  }
  const long long t1 = clock64();
  out[lane * 3] = X, out[lane * 3 + 1] = W1, out[lane * 3 + 2] = W2;
  if (lane == 0) cyc[blockIdx.x] = (t1 - t0) / reps;
}

// the accumulation block of k_persistent_pv<.., RIPPLE, ..> as shipped: masks for shifts 1..7, one mask from shift 8 on (a vertex
// of more than 8 edges has its row to itself), idle lanes disabled as sources (a DPP read of a disabled lane leaves the
// destination untouched: bound_ctrl 0) -- their contributions are set to garbage here to prove it
__global__ void __launch_bounds__(64) k_ripple_ship(const int* first, const int* deg, const int* active, const float* c5, float* out, long long* cyc, int reps, int maxdeg) {
  const int lane = threadIdx.x;
  const int f = first[lane], d = deg[lane];
  const bool head = f == lane;
  const unsigned degx = head ? (active[lane] ? (unsigned)d : 0u) : (active[lane] ? 255u : 0u);
  float cx = c5[lane * 5], a1 = c5[lane * 5 + 1], a2 = c5[lane * 5 + 2], b1 = c5[lane * 5 + 3], b2 = c5[lane * 5 + 4];
  if (!active[lane]) cx = a1 = a2 = b1 = b2 = 1e30f;
  float x = 1.0f + lane, w1 = 0.5f, w2 = -0.25f;
  float X = 0, W1 = 0, W2 = 0;
  const unsigned long long m1 = __ballot(degx > 1u), m2 = __ballot(degx > 2u), m3 = __ballot(degx > 3u), m4 = __ballot(degx > 4u), m5 = __ballot(degx > 5u),
                           m6 = __ballot(degx > 6u), m7 = __ballot(degx > 7u), m8 = __ballot(degx > 8u);
  const long long t0 = clock64();
  for (int r = 0; r < reps; ++r) {
    X = x + cx, W1 = (w1 + a1) + b1, W2 = (w2 + a2) + b2;
#define RM(J, M) "s_mov_b64 exec, %[" #M "]\n\t" PV_ADDS(J)
    asm volatile("s_nop 1\n\t" RM(1, m1) RM(2, m2) RM(3, m3) RM(4, m4) RM(5, m5) RM(6, m6) RM(7, m7)
                 "s_cmp_le_u32 %[md], 8\n\ts_cbranch_scc1 9f\n\t" RM(8, m8) PV_ADDS(9)
                 "s_cmp_le_u32 %[md], 10\n\ts_cbranch_scc1 9f\n\t" PV_ADDS(10) PV_ADDS(11)
                 "s_cmp_le_u32 %[md], 12\n\ts_cbranch_scc1 9f\n\t" PV_ADDS(12) PV_ADDS(13) PV_ADDS(14) PV_ADDS(15)
                 "9:\n\ts_mov_b64 exec, -1"
                 : [X] "+v"(X), [W1] "+v"(W1), [W2] "+v"(W2)
                 : [cx] "v"(cx), [a1] "v"(a1), [a2] "v"(a2), [b1] "v"(b1), [b2] "v"(b2), [md] "s"(maxdeg), [m1] "s"(m1), [m2] "s"(m2), [m3] "s"(m3),
                   [m4] "s"(m4), [m5] "s"(m5), [m6] "s"(m6), [m7] "s"(m7), [m8] "s"(m8) : "scc");
#undef RM
    x = head ? X * 1e-30f + x : x;
  }
  const long long t1 = clock64();
  out[lane * 3] = X, out[lane * 3 + 1] = W1, out[lane * 3 + 2] = W2;
  if (lane == 0) cyc[blockIdx.x] = (t1 - t0) / reps;
}

template <int J>
__device__ __forceinline__ void ripple_step(float& X, float& W1, float& W2, const float cx, const float a1, const float a2, const float b1,
                                            const float b2, const unsigned degx) {
  asm volatile("v_cmpx_gt_u32_e32 %[d], %[j]\n\t"
               "v_add_f32_dpp %[X], %[cx], %[X] row_shl:%[sh] row_mask:0xf bank_mask:0xf\n\t"
               "v_add_f32_dpp %[W1], %[a1], %[W1] row_shl:%[sh] row_mask:0xf bank_mask:0xf\n\t"
               "v_add_f32_dpp %[W2], %[a2], %[W2] row_shl:%[sh] row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 0\n\t"
               "v_add_f32_dpp %[W1], %[b1], %[W1] row_shl:%[sh] row_mask:0xf bank_mask:0xf\n\t"
               "v_add_f32_dpp %[W2], %[b2], %[W2] row_shl:%[sh] row_mask:0xf bank_mask:0xf\n\t"
               : [X] "+v"(X), [W1] "+v"(W1), [W2] "+v"(W2)
               : [cx] "v"(cx), [a1] "v"(a1), [a2] "v"(a2), [b1] "v"(b1), [b2] "v"(b2), [d] "v"(degx), [j] "v"((unsigned)J), [sh] "n"(J)
               : "vcc", "exec");
}

__global__ void __launch_bounds__(64) k_ripple(const int* first, const int* deg, const float* c5, float* out, long long* cyc, int reps, int maxdeg) {
  const int lane = threadIdx.x;
  const int f = first[lane], d = deg[lane];
  const bool head = f == lane;
  const unsigned degx = head ? (unsigned)d : 255u;
  float cx = c5[lane * 5], a1 = c5[lane * 5 + 1], a2 = c5[lane * 5 + 2], b1 = c5[lane * 5 + 3], b2 = c5[lane * 5 + 4];
  float x = 1.0f + lane, w1 = 0.5f, w2 = -0.25f;
  float X = 0, W1 = 0, W2 = 0;
  const long long t0 = clock64();
  for (int r = 0; r < reps; ++r) {
    X = x + cx, W1 = (w1 + a1) + b1, W2 = (w2 + a2) + b2;
    asm volatile("" : "+v"(X), "+v"(W1), "+v"(W2));
    if (maxdeg > 1) ripple_step<1>(X, W1, W2, cx, a1, a2, b1, b2, degx);
    if (maxdeg > 2) ripple_step<2>(X, W1, W2, cx, a1, a2, b1, b2, degx);
    if (maxdeg > 3) ripple_step<3>(X, W1, W2, cx, a1, a2, b1, b2, degx);
    if (maxdeg > 4) ripple_step<4>(X, W1, W2, cx, a1, a2, b1, b2, degx);
    if (maxdeg > 5) ripple_step<5>(X, W1, W2, cx, a1, a2, b1, b2, degx);
    if (maxdeg > 6) ripple_step<6>(X, W1, W2, cx, a1, a2, b1, b2, degx);
    if (maxdeg > 7) ripple_step<7>(X, W1, W2, cx, a1, a2, b1, b2, degx);
    if (maxdeg > 8) {
      ripple_step<8>(X, W1, W2, cx, a1, a2, b1, b2, degx);
      if (maxdeg > 9) ripple_step<9>(X, W1, W2, cx, a1, a2, b1, b2, degx);
      if (maxdeg > 10) ripple_step<10>(X, W1, W2, cx, a1, a2, b1, b2, degx);
      if (maxdeg > 11) ripple_step<11>(X, W1, W2, cx, a1, a2, b1, b2, degx);
      if (maxdeg > 12) ripple_step<12>(X, W1, W2, cx, a1, a2, b1, b2, degx);
      if (maxdeg > 13) ripple_step<13>(X, W1, W2, cx, a1, a2, b1, b2, degx);
      if (maxdeg > 14) ripple_step<14>(X, W1, W2, cx, a1, a2, b1, b2, degx);
      if (maxdeg > 15) ripple_step<15>(X, W1, W2, cx, a1, a2, b1, b2, degx);
    }
    asm volatile("s_mov_b64 exec, -1" ::: "exec");
    // dependency into the next repetition (as the real step has: the next dual update uses the new state)
    x = head ? X * 1e-30f + x : x;
  }
  const long long t1 = clock64();
  out[lane * 3] = X, out[lane * 3 + 1] = W1, out[lane * 3 + 2] = W2;
  if (lane == 0) cyc[blockIdx.x] = (t1 - t0) / reps;
}

// the current scheme: contributions to an LDS slab, every lane of a vertex reads its vertex's `stride` slots back in order
__global__ void __launch_bounds__(64) k_slab(const int* first, const int* deg, const int* vtx, const float* c5, float* out, long long* cyc, int reps, int stride) {
  __shared__ float4 slabA[64 * 17];
  __shared__ float slabC[64 * 16];
  const int lane = threadIdx.x;
  const int f = first[lane], v = vtx[lane], pos = lane - f;
  float cx = c5[lane * 5], a1 = c5[lane * 5 + 1], a2 = c5[lane * 5 + 2], b1 = c5[lane * 5 + 3], b2 = c5[lane * 5 + 4];
  float x = 1.0f + f, w1 = 0.5f, w2 = -0.25f;
  for (int i = lane; i < 64 * 17; i += 64) slabA[i] = make_float4(-0.f, -0.f, -0.f, -0.f);
  for (int i = lane; i < 64 * 16; i += 64) slabC[i] = -0.f;
  __syncthreads();
  float X = 0, W1 = 0, W2 = 0;
  const int sA = stride + 1;
  const long long t0 = clock64();
  for (int r = 0; r < reps; ++r) {
    slabA[v * sA + pos] = make_float4(a1, a2, b1, b2);
    slabC[v * stride + pos] = cx;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    X = x, W1 = w1, W2 = w2;
    for (int k0 = 0; k0 < stride; k0 += 4) {
      float4 c[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) c[k] = slabA[v * sA + k0 + k];
      const float4 cc = *reinterpret_cast<const float4*>(&slabC[v * stride + k0]);
      const float cxs[4] = {cc.x, cc.y, cc.z, cc.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        X = X + cxs[k];
        W1 = (W1 + c[k].x) + c[k].z;
        W2 = (W2 + c[k].y) + c[k].w;
      }
    }
    x = X * 1e-30f + x;
    asm volatile("" : "+v"(x));
  }
  const long long t1 = clock64();
  out[lane * 3] = X, out[lane * 3 + 1] = W1, out[lane * 3 + 2] = W2;
  if (lane == 0) cyc[blockIdx.x] = (t1 - t0) / reps;
}

int main() {
  // segments: rows of 16 lanes; degrees chosen to fill rows: (6,6,4) (7,5,4) (16) (8,8)  -- second config caps at 8
  for (int cfg = 0; cfg < 3; ++cfg) {
    std::vector<int> degs = cfg == 0 ? std::vector<int>{6, 6, 4, 7, 5, 4, 16, 8, 8} : cfg == 1 ? std::vector<int>{6, 6, 4, 7, 5, 4, 6, 5, 5, 8, 8} : std::vector<int>{9, 6, 5, 13, 7, 6};
    std::vector<int> first(64, 0), deg(64, 1), vtx(64, 0), act(64, 0);
    int lane = 0, v = 0, maxdeg = 1;
    std::vector<int> vfirst;
    for (int d : degs) {
      if (lane % 16 + d > 16 || (cfg == 2 && (d > 8 || (lane % 16 && deg[lane - 1] > 8)))) { while (lane % 16) { first[lane] = lane, deg[lane] = 1, vtx[lane] = 63; ++lane; } }
      vfirst.push_back(lane);
      for (int k = 0; k < d; ++k) first[lane + k] = lane, deg[lane + k] = d, vtx[lane + k] = v, act[lane + k] = 1;
      lane += d, ++v;
      maxdeg = std::max(maxdeg, d);
    }
    for (; lane < 64; ++lane) first[lane] = lane, deg[lane] = 1, vtx[lane] = 63;
    std::vector<float> c5(64 * 5);
    unsigned s = 12345;
    for (auto& f : c5) { s = s * 1664525u + 1013904223u; f = ((int)(s >> 8) % 2000 - 1000) * 1.37e-3f; }
    int *dfirst, *ddeg, *dvtx; float *dc, *dout; long long* dcyc;
    CHECK(hipMalloc(&dfirst, 256)); CHECK(hipMalloc(&ddeg, 256)); CHECK(hipMalloc(&dvtx, 256)); CHECK(hipMalloc(&dc, 64 * 5 * 4));
    CHECK(hipMalloc(&dout, 64 * 3 * 4)); CHECK(hipMalloc(&dcyc, 1024 * 8));
    CHECK(hipMemcpy(dfirst, first.data(), 256, hipMemcpyHostToDevice)); CHECK(hipMemcpy(ddeg, deg.data(), 256, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dvtx, vtx.data(), 256, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dc, c5.data(), 64 * 5 * 4, hipMemcpyHostToDevice));
    // expected: sequential sums per vertex
    std::vector<float> ex(64 * 3, 0.f);
    for (size_t i = 0; i < degs.size(); ++i) {
      const int f0 = vfirst[i];
      float X = 1.0f + f0, W1 = 0.5f, W2 = -0.25f;
      for (int k = 0; k < degs[i]; ++k) { const float* c = &c5[(f0 + k) * 5]; X = X + c[0]; W1 = (W1 + c[1]) + c[3]; W2 = (W2 + c[2]) + c[4]; }
      ex[f0 * 3] = X, ex[f0 * 3 + 1] = W1, ex[f0 * 3 + 2] = W2;
    }
    std::vector<float> out(64 * 3); long long cyc[1024];
    {
      int* dact; CHECK(hipMalloc(&dact, 256)); CHECK(hipMemcpy(dact, act.data(), 256, hipMemcpyHostToDevice));
      hipLaunchKernelGGL(k_ripple_ship, dim3(1024), dim3(64), 0, 0, dfirst, ddeg, dact, dc, dout, dcyc, 2000, maxdeg);
      CHECK(hipDeviceSynchronize());
      CHECK(hipMemcpy(out.data(), dout, 64 * 3 * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(cyc, dcyc, sizeof(long long) * 1024, hipMemcpyDeviceToHost));
      int bad = 0;
      for (size_t i = 0; i < degs.size(); ++i) for (int k = 0; k < 3; ++k) bad += std::memcmp(&out[vfirst[i] * 3 + k], &ex[vfirst[i] * 3 + k], 4) != 0;
      printf("cfg %d maxdeg %2d as shipped (idle lanes hold garbage, disabled as sources): %lld cycles per accumulation, mismatching head sums %d\n", cfg, maxdeg, cyc[0], bad);
    }
    if (cfg < 2)
    for (int nb : {1, 1024}) {
      hipLaunchKernelGGL(k_ripple, dim3(nb), dim3(64), 0, 0, dfirst, ddeg, dc, dout, dcyc, 2000, maxdeg);
      CHECK(hipDeviceSynchronize());
      CHECK(hipMemcpy(out.data(), dout, 64 * 3 * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(cyc, dcyc, sizeof(long long) * nb, hipMemcpyDeviceToHost));
      int bad = 0;
      for (size_t i = 0; i < degs.size(); ++i) for (int k = 0; k < 3; ++k) bad += std::memcmp(&out[vfirst[i] * 3 + k], &ex[vfirst[i] * 3 + k], 4) != 0;
      printf("cfg %d maxdeg %2d ripple  blocks %4d: %lld cycles per accumulation, mismatching head sums %d\n", cfg, maxdeg, nb, cyc[0], bad);
      if (cfg == 1) {
        for (int mode = 1; mode <= 6; ++mode) {
          if (mode == 1) hipLaunchKernelGGL(k_ripple8<1>, dim3(nb), dim3(64), 0, 0, dfirst, ddeg, dc, dout, dcyc, 2000);
          if (mode == 2) hipLaunchKernelGGL(k_ripple8<2>, dim3(nb), dim3(64), 0, 0, dfirst, ddeg, dc, dout, dcyc, 2000);
          if (mode == 3) hipLaunchKernelGGL(k_ripple8<3>, dim3(nb), dim3(64), 0, 0, dfirst, ddeg, dc, dout, dcyc, 2000);
          if (mode == 4) hipLaunchKernelGGL(k_ripple_k<4>, dim3(nb), dim3(64), 0, 0, dfirst, ddeg, dc, dout, dcyc, 2000, 8);
          if (mode == 5) hipLaunchKernelGGL(k_ripple_k<5>, dim3(nb), dim3(64), 0, 0, dfirst, ddeg, dc, dout, dcyc, 2000, 8);
          if (mode == 6) hipLaunchKernelGGL(k_ripple_k<6>, dim3(nb), dim3(64), 0, 0, dfirst, ddeg, dc, dout, dcyc, 2000, 8);
          CHECK(hipDeviceSynchronize());
          CHECK(hipMemcpy(out.data(), dout, 64 * 3 * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(cyc, dcyc, sizeof(long long) * nb, hipMemcpyDeviceToHost));
          bad = 0;
          for (size_t i = 0; i < degs.size(); ++i) for (int k = 0; k < 3; ++k) bad += std::memcmp(&out[vfirst[i] * 3 + k], &ex[vfirst[i] * 3 + k], 4) != 0;
          printf("   mode %d (1 sgpr masks, 2 no exec changes, 3 plain adds, 4 kernel block: v_cmp ahead + branches, 5 same without branches, 6 sgpr masks + branches): %lld cycles, mismatches %d\n", mode, cyc[0], bad);
        }
      }
      const int stride = std::max(8, (maxdeg + 3) & ~3);
      hipLaunchKernelGGL(k_slab, dim3(nb), dim3(64), 0, 0, dfirst, ddeg, dvtx, dc, dout, dcyc, 2000, stride);
      CHECK(hipDeviceSynchronize());
      CHECK(hipMemcpy(out.data(), dout, 64 * 3 * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(cyc, dcyc, sizeof(long long) * nb, hipMemcpyDeviceToHost));
      bad = 0;
      for (size_t i = 0; i < degs.size(); ++i) for (int k = 0; k < 3; ++k) bad += std::memcmp(&out[vfirst[i] * 3 + k], &ex[vfirst[i] * 3 + k], 4) != 0;
      printf("cfg %d stride %2d slab    blocks %4d: %lld cycles per accumulation, mismatching head sums %d\n", cfg, stride, nb, cyc[0], bad);
    }
  }
  return 0;
}
