// Development probe: per-instruction latencies of one wavefront on gfx950 (dependent chains), shader-clock cycles.
// hipcc --offload-arch=gfx950 -O3 -o latency latency.hip && ./latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define N 256
__global__ void probe(unsigned long long *out, int *sink, int seed) {
  __shared__ int lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (i * 7 + 1) & 1023;
  __syncthreads();
  unsigned long long t0, t1;
  int lane = threadIdx.x;
  // (a) dependent LDS read chain (uniform address)
  int idx = seed & 1023;
  t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int i = 0; i < N; i++) idx = lds[idx];
  t1 = __builtin_readcyclecounter();
  if (lane == 0) out[0] = t1 - t0;
  sink[lane] = idx;
  // (b) dependent VALU chain
  int v = seed + lane;
  t0 = __builtin_readcyclecounter();
#pragma unroll
  for (int i = 0; i < N; i++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v) : "v"(lane));
  t1 = __builtin_readcyclecounter();
  if (lane == 0) out[1] = t1 - t0;
  sink[64 + lane] = v;
  // (c) dependent SALU chain
  int s = seed;
  t0 = __builtin_readcyclecounter();
#pragma unroll
  for (int i = 0; i < N; i++) asm volatile("s_add_u32 %0, %0, 3" : "+s"(s) : : "scc");
  t1 = __builtin_readcyclecounter();
  if (lane == 0) out[2] = t1 - t0;
  sink[128 + lane] = s;
  // (d) VALU -> readfirstlane -> SALU -> VALU round trip
  v = seed + lane;
  t0 = __builtin_readcyclecounter();
#pragma unroll
  for (int i = 0; i < N; i++) {
    int ss;
    asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(ss) : "v"(v));
    asm volatile("s_add_u32 %0, %0, 3" : "+s"(ss) : : "scc");
    asm volatile("v_add_u32 %0, %1, %2" : "=v"(v) : "s"(ss), "v"(lane));
  }
  t1 = __builtin_readcyclecounter();
  if (lane == 0) out[3] = t1 - t0;
  sink[192 + lane] = v;
  // (e) v_cmp -> s_and_b64 -> v_cndmask chain
  v = seed + lane;
  t0 = __builtin_readcyclecounter();
#pragma unroll
  for (int i = 0; i < N; i++) {
    asm volatile("v_cmp_gt_u32 vcc, %1, %2\n\ts_and_b64 vcc, vcc, exec\n\tv_cndmask_b32 %0, %1, %2, vcc" : "=v"(v) : "v"(v), "v"(lane) : "vcc");
  }
  t1 = __builtin_readcyclecounter();
  if (lane == 0) out[4] = t1 - t0;
  sink[256 + lane] = v;
  // (f) taken scalar branches
  s = seed;
  t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int i = 0; i < N; i++) {
    asm volatile("s_add_u32 %0, %0, 1\n\ts_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 1f\n\ts_branch 2f\n1:\n\ts_add_u32 %0, %0, 5\n2:\n\ts_nop 0" : "+s"(s) : : "scc");
  }
  t1 = __builtin_readcyclecounter();
  if (lane == 0) out[5] = t1 - t0;
  sink[320 + lane] = s;
  // (g) LDS read -> readfirstlane -> address chain (what the search does)
  idx = seed & 1023;
  t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int i = 0; i < N; i++) {
    int r = lds[idx];
    idx = __builtin_amdgcn_readfirstlane(r);
  }
  t1 = __builtin_readcyclecounter();
  if (lane == 0) out[6] = t1 - t0;
  sink[384 + lane] = idx;
  // (i) four independent VALU chains, interleaved
  {
    int a0 = seed + lane, a1 = seed - lane, a2 = seed ^ lane, a3 = seed * 3 + lane;
    t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < N / 4; i++)
      asm volatile("v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(lane));
    t1 = __builtin_readcyclecounter();
    if (lane == 0) out[8] = t1 - t0;
    sink[512 + lane] = a0 + a1 + a2 + a3;
  }
  // (j) independent SALU and VALU chains, interleaved 1:1
  {
    int a0 = seed + lane, s0 = seed;
    t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < N / 2; i++) asm volatile("v_add_u32 %0, %0, %2\n\ts_add_u32 %1, %1, 3" : "+v"(a0), "+s"(s0) : "v"(lane) : "scc");
    t1 = __builtin_readcyclecounter();
    if (lane == 0) out[9] = t1 - t0;
    sink[576 + lane] = a0 + s0;
  }
  // (k) two independent SALU chains
  {
    int s0 = seed, s1 = seed + 1;
    t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < N / 2; i++) asm volatile("s_add_u32 %0, %0, 3\n\ts_add_u32 %1, %1, 5" : "+s"(s0), "+s"(s1) : : "scc");
    t1 = __builtin_readcyclecounter();
    if (lane == 0) out[10] = t1 - t0;
    sink[640 + lane] = s0 + s1;
  }
  // (h) ballot -> ctz -> readlane chain
  v = seed + lane;
  t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int i = 0; i < N; i++) {
    unsigned long long b = __builtin_amdgcn_ballot_w64((v & 1) != 0);
    int l = b ? __builtin_ctzll(b) : 0;
    v += __builtin_amdgcn_readlane(v, l) | 1;
  }
  t1 = __builtin_readcyclecounter();
  if (lane == 0) out[7] = t1 - t0;
  sink[448 + lane] = v;
}

int main() {
  unsigned long long *out;
  int *sink;
  (void)hipMalloc(&out, 128);
  (void)hipMalloc(&sink, 4096);
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, out, sink, 5 + rep);
    (void)hipDeviceSynchronize();
  }
  unsigned long long h[11];
  (void)hipMemcpy(h, out, 88, hipMemcpyDeviceToHost);
  const char *names[11] = {"LDS read chain (loop)", "VALU add chain", "SALU add chain", "readfirstlane+SALU+VALU (3 instr)", "v_cmp+s_and+v_cndmask (3 instr)",
                          "loop with 1 taken + 1 untaken branch (6 instr)", "LDS read -> readfirstlane chain", "ballot->ctz->readlane->add chain",
                          "4 independent VALU chains (per instruction, x N)", "VALU + SALU independent (per instruction)", "2 independent SALU chains (per instruction)"};
  for (int i = 0; i < 11; i++) printf("%-50s %7.1f cycles per iteration\n", names[i], (double)h[i] / N);
  return 0;
}
