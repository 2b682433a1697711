// Read-bandwidth probe for candidate tile->wave mappings of the stream kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ uint32_t mix(uint4 v) { return v.x ^ v.y ^ v.z ^ v.w; }

// A: chunk-interleaved grid stride: wave-iteration i of global wave w reads KiB (i * nwaves + w)
__global__ __launch_bounds__(512) void kA(const uint4* __restrict__ t, uint64_t nkib, uint32_t* out) {
  uint32_t lane = threadIdx.x & 63; uint64_t w = (uint64_t)blockIdx.x * 8 + (threadIdx.x >> 6), nw = (uint64_t)gridDim.x * 8;
  uint32_t acc = 0;
  for (uint64_t k = w; k < nkib; k += nw) acc ^= mix(t[k * 64 + lane]);
  if (acc == 0x12345) out[0] = acc;
}
// B: TILE KiB contiguous per wave, DEPTH loads in flight
template <int TILE, int DEPTH>
__global__ __launch_bounds__(512) void kB(const uint4* __restrict__ t, uint64_t ntiles, uint32_t* out) {
  uint32_t lane = threadIdx.x & 63; uint64_t w = (uint64_t)blockIdx.x * 8 + (threadIdx.x >> 6), nw = (uint64_t)gridDim.x * 8;
  uint32_t acc = 0;
  for (uint64_t tile = w; tile < ntiles; tile += nw) {
    const uint4* p = t + tile * (TILE * 64) + lane;
    uint4 buf[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; d++) buf[d] = p[d * 64];
#pragma unroll
    for (int it = 0; it < TILE; it++) {
      uint4 cur = buf[it % DEPTH];
      if (it + DEPTH < TILE) buf[it % DEPTH] = p[(it + DEPTH) * 64];
      acc ^= mix(cur);
    }
  }
  if (acc == 0x12345) out[0] = acc;
}
// C: like B but consecutive tiles are dealt to consecutive ITERATION slots of the whole grid:
// WG g owns a contiguous region of 8*TILE KiB; in iteration i its 8 waves read 8 consecutive KiB (i*8 + wave)
template <int TILE>
__global__ __launch_bounds__(512) void kC(const uint4* __restrict__ t, uint64_t nregions, uint32_t* out) {
  uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t acc = 0;
  for (uint64_t r = blockIdx.x; r < nregions; r += gridDim.x) {
    const uint4* p = t + r * (uint64_t)(8 * TILE * 64) + wave * 64 + lane;
    uint4 buf[3];
#pragma unroll
    for (int d = 0; d < 3; d++) buf[d] = p[d * 8 * 64];
#pragma unroll
    for (int it = 0; it < TILE; it++) {
      uint4 cur = buf[it % 3];
      if (it + 3 < TILE) buf[it % 3] = p[(it + 3) * 8 * 64];
      acc ^= mix(cur);
    }
  }
  if (acc == 0x12345) out[0] = acc;
}

template <typename F> void run(const char* name, uint64_t bytes, F f) {
  hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  f(); CHECK(hipDeviceSynchronize());
  float best = 1e9;
  for (int r = 0; r < 3; r++) { CHECK(hipEventRecord(a)); f(); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b)); float ms; CHECK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms; }
  printf("%-34s %8.3f ms  %8.1f GB/s\n", name, best, bytes / best / 1e6);
}
int main(int argc, char** argv) {
  uint64_t bytes = (argc > 1 ? atoll(argv[1]) : 4ull) << 30;
  uint4* t; uint32_t* out; CHECK(hipMalloc(&t, bytes)); CHECK(hipMalloc(&out, 64)); CHECK(hipMemset(t, 1, bytes));
  uint64_t nkib = bytes >> 10;
  for (int wgs : {512, 1024, 2048}) {
    printf("grid %d WGs x 512 threads\n", wgs);
    run("A interleaved 1KiB", bytes, [&] { kA<<<wgs, 512>>>(t, nkib, out); });
    run("B tile16 depth3", bytes, [&] { kB<16, 3><<<wgs, 512>>>(t, nkib / 16, out); });
    run("B tile16 depth6", bytes, [&] { kB<16, 6><<<wgs, 512>>>(t, nkib / 16, out); });
    run("B tile16 depth16", bytes, [&] { kB<16, 16><<<wgs, 512>>>(t, nkib / 16, out); });
    run("B tile4 depth4", bytes, [&] { kB<4, 4><<<wgs, 512>>>(t, nkib / 4, out); });
    run("B tile64 depth4", bytes, [&] { kB<64, 4><<<wgs, 512>>>(t, nkib / 64, out); });
    run("C region128K (8 waves x 16)", bytes, [&] { kC<16><<<wgs, 512>>>(t, nkib / 128, out); });
  }
  return 0;
}
