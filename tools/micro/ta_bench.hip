// ta_bench.hip -- how much does a divergent global_load_dwordx4 cost on MI355X as a function of active lanes and of how
// many lanes share a 128-byte line?  Every lane chases `steps` dependent 16-byte loads through a table that fits in L2
// (2 MiB, like the BVH of the benchmark scene).  Build: hipcc --offload-arch=gfx950 -O3 ta_bench.hip -o ta_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>

// active: lanes per wave that take part (others idle at the loop).  group: consecutive lanes that read the same 128-B line
// (each its own 16-B piece when group <= 8).  loadsPerStep: 16-B loads issued back to back per step by each lane (1..8)
template <int LOADS>
__global__ void __launch_bounds__(128) chase(const float4* __restrict__ table, int lines, int steps, int active, int group, unsigned* out) {
  const int lane = threadIdx.x & 63;
  const int gtid = blockIdx.x * blockDim.x + threadIdx.x;
  if (lane >= active) return;
  unsigned key = (unsigned)(gtid / group) * 2654435761u + 12345u;
  const int piece = (group <= 8) ? (lane % group) : 0;
  float acc = 0.0f;
  for (int s = 0; s < steps; s++) {
    const unsigned line = (key >> 8) % (unsigned)lines;
    const float4* p = table + (size_t)line * 8;
    float4 v[LOADS];
#pragma unroll
    for (int k = 0; k < LOADS; k++) v[k] = p[(piece + k) & 7];
    float sum = 0.0f;
#pragma unroll
    for (int k = 0; k < LOADS; k++) sum += v[k].x + v[k].w;
    acc += sum;
    key = key * 1664525u + 1013904223u + (unsigned)(int)(sum * 0.0f);   // dependent on the load
  }
  out[gtid] = __float_as_uint(acc) + key;
}

// octet-cooperative reads through a raw buffer: `liveOctets` of the 8 octets of a wave read one line each (8 x 16 B),
// the other octets either pass an out-of-range offset (oob = 1: hardware range check) or are switched off (oob = 0)
__global__ void __launch_bounds__(128) chase_octets(const float4* __restrict__ table, int lines, int steps, int liveOctets, int oob, unsigned* out) {
  const int lane = threadIdx.x & 63;
  const int gtid = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = (lane >> 3) < liveOctets;
  if (!live && !oob) return;
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(table), 0, lines * 128, 0x00020000);
  unsigned key = (unsigned)(gtid / 8) * 2654435761u + 12345u;
  typedef unsigned int v4u __attribute__((ext_vector_type(4)));
  unsigned acc = 0;
  for (int s = 0; s < steps; s++) {
    const unsigned line = (key >> 8) % (unsigned)lines;
    const unsigned off = live ? line * 128u + (lane & 7) * 16u : 0xFFFFFF00u;
    const v4u v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
    acc += v.x;
    key = key * 1664525u + 1013904223u + (v.w & 0u);
  }
  out[gtid] = acc + key;
}
// fully divergent reads through a raw buffer (32-bit offsets) for comparison with the 64-bit global_load form of chase<>
template <int LOADS>
__global__ void __launch_bounds__(128) chase_buf(const float4* __restrict__ table, int lines, int steps, int active, unsigned* out) {
  const int lane = threadIdx.x & 63;
  const int gtid = blockIdx.x * blockDim.x + threadIdx.x;
  if (lane >= active) return;
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(table), 0, lines * 128, 0x00020000);
  unsigned key = (unsigned)gtid * 2654435761u + 12345u;
  typedef unsigned int v4u __attribute__((ext_vector_type(4)));
  unsigned acc = 0;
  for (int s = 0; s < steps; s++) {
    const unsigned line = (key >> 8) % (unsigned)lines;
    v4u v[LOADS];
#pragma unroll
    for (int k = 0; k < LOADS; k++) v[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, line * 128u + k * 16u, 0, 0);
    unsigned sum = 0;
#pragma unroll
    for (int k = 0; k < LOADS; k++) sum += v[k].x;
    acc += sum;
    key = key * 1664525u + 1013904223u + (sum & 0u);
  }
  out[gtid] = acc + key;
}
template <int LOADS>
static void run_buf(const float4* table, int lines, unsigned* out, int blocks, int active, int steps) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(chase_buf<LOADS>, dim3(blocks), dim3(128), 0, 0, table, lines, steps, active, out);
  hipEventRecord(a);
  hipLaunchKernelGGL(chase_buf<LOADS>, dim3(blocks), dim3(128), 0, 0, table, lines, steps, active, out);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double waveInstr = double(blocks) * 2 * steps * LOADS;
  printf("buffer loads/step %d active %2d divergent: %7.3f ms  %6.1f clk/wave-instr/CU  %6.2f clk/lane-load/CU\n", LOADS, active, ms,
         ms * 1e-3 * 2.4e9 / (waveInstr / 256.0), ms * 1e-3 * 2.4e9 / (waveInstr * active / 256.0));
  hipEventDestroy(a); hipEventDestroy(b);
}

static void run_octets(const float4* table, int lines, unsigned* out, int blocks, int liveOctets, int oob, int steps) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(chase_octets, dim3(blocks), dim3(128), 0, 0, table, lines, steps, liveOctets, oob, out);
  hipEventRecord(a);
  hipLaunchKernelGGL(chase_octets, dim3(blocks), dim3(128), 0, 0, table, lines, steps, liveOctets, oob, out);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double waveInstr = double(blocks) * 2 * steps;
  printf("octet reads, %d live octets, others %s: %7.3f ms  %6.1f clk/wave-instr/CU  %6.2f clk/line/CU\n", liveOctets, oob ? "out of range" : "masked off  ", ms,
         ms * 1e-3 * 2.4e9 / (waveInstr / 256.0), ms * 1e-3 * 2.4e9 / (waveInstr * liveOctets / 256.0));
  hipEventDestroy(a); hipEventDestroy(b);
}

// same as chase<> with group 1, but the LOADS pieces are addressed as base + immediate offset (global_load ... offset:N)
template <int LOADS>
__global__ void __launch_bounds__(128) chase_imm(const float4* __restrict__ table, int lines, int steps, int active, unsigned* out) {
  const int lane = threadIdx.x & 63;
  const int gtid = blockIdx.x * blockDim.x + threadIdx.x;
  if (lane >= active) return;
  unsigned key = (unsigned)gtid * 2654435761u + 12345u;
  float acc = 0.0f;
  for (int s = 0; s < steps; s++) {
    const unsigned line = (key >> 8) % (unsigned)lines;
    const float4* p = table + (size_t)line * 8;
    float4 v[LOADS];
#pragma unroll
    for (int k = 0; k < LOADS; k++) v[k] = p[k];
    float sum = 0.0f;
#pragma unroll
    for (int k = 0; k < LOADS; k++) sum += v[k].x + v[k].w;
    acc += sum;
    key = key * 1664525u + 1013904223u + (unsigned)(int)(sum * 0.0f);
  }
  out[gtid] = __float_as_uint(acc) + key;
}
template <int LOADS>
static void run_imm(const float4* table, int lines, unsigned* out, int blocks, int active, int steps) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(chase_imm<LOADS>, dim3(blocks), dim3(128), 0, 0, table, lines, steps, active, out);
  hipEventRecord(a);
  hipLaunchKernelGGL(chase_imm<LOADS>, dim3(blocks), dim3(128), 0, 0, table, lines, steps, active, out);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double waveInstr = double(blocks) * 2 * steps * LOADS;
  printf("global loads/step %d active %2d divergent, base + immediate: %7.3f ms  %6.1f clk/wave-instr/CU  %6.2f clk/lane-load/CU\n", LOADS, active, ms,
         ms * 1e-3 * 2.4e9 / (waveInstr / 256.0), ms * 1e-3 * 2.4e9 / (waveInstr * active / 256.0));
  hipEventDestroy(a); hipEventDestroy(b);
}

template <int LOADS>
static void run(const float4* table, int lines, unsigned* out, int blocks, int active, int group, int steps) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(chase<LOADS>, dim3(blocks), dim3(128), 0, 0, table, lines, steps, active, group, out);
  hipEventRecord(a);
  hipLaunchKernelGGL(chase<LOADS>, dim3(blocks), dim3(128), 0, 0, table, lines, steps, active, group, out);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double waveInstr = double(blocks) * 2 * steps * LOADS;
  const double laneLoads = waveInstr * active;
  printf("loads/step %d active %2d group %2d: %7.3f ms  %6.1f clk/wave-instr/CU  %6.2f clk/lane-load/CU  %7.1f GB/s useful\n", LOADS, active, group, ms,
         ms * 1e-3 * 2.4e9 / (waveInstr / 256.0), ms * 1e-3 * 2.4e9 / (laneLoads / 256.0), laneLoads * 16.0 / ms / 1e6);
  hipEventDestroy(a); hipEventDestroy(b);
}

int main(int argc, char** argv) {
  const int lines = (argc > 1 ? atoi(argv[1]) : 16384);   // 16384 lines * 128 B = 2 MiB
  const int steps = 256;
  const int blocks = 256 * 10;                            // 10 blocks of 2 waves per CU
  float4* table; unsigned* out;
  hipMalloc(&table, size_t(lines) * 128);
  hipMalloc(&out, size_t(blocks) * 128 * 4);
  std::vector<float> h(size_t(lines) * 32, 1.0f);
  hipMemcpy(table, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  printf("table %d lines (%.1f MiB), %d blocks x 128 threads, %d dependent steps\n", lines, lines * 128.0 / 1048576.0, blocks, steps);
  for (int active : {64, 32, 16, 8}) run<1>(table, lines, out, blocks, active, 1, steps);
  for (int group : {1, 2, 4, 8}) run<1>(table, lines, out, blocks, 64, group, steps);
  for (int active : {64, 32, 16}) run<8>(table, lines, out, blocks, active, 1, steps);
  for (int group : {2, 4, 8}) run<2>(table, lines, out, blocks, 64, group, steps);
  run<8>(table, lines, out, blocks, 64, 64, steps);      // whole wave reads the same line (broadcast)
  for (int active : {64, 16}) { run_imm<8>(table, lines, out, blocks, active, steps); run_imm<2>(table, lines, out, blocks, active, steps); }
  for (int active : {64, 16}) { run_buf<1>(table, lines, out, blocks, active, steps); run_buf<8>(table, lines, out, blocks, active, steps); }
  for (int oob : {0, 1}) for (int live : {8, 4, 2, 1}) run_octets(table, lines, out, blocks, live, oob, steps);
  return 0;
}
