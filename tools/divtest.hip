// Checks 64-bit integer division / modulo on device against the host for random operands.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
__global__ void k(const uint64_t* a, const uint32_t* b, const uint64_t* c, uint64_t* q1, uint64_t* r1, uint64_t* q2, uint64_t* r2, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  q1[i] = a[i] / b[i]; r1[i] = a[i] % b[i];   // 64 by 32-bit variable
  q2[i] = a[i] / c[i]; r2[i] = a[i] % c[i];   // 64 by 64
}
int main() {
  int n = 1 << 20;
  std::vector<uint64_t> a(n), c(n), q1(n), r1(n), q2(n), r2(n);
  std::vector<uint32_t> b(n);
  uint64_t s = 88172645463325252ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
  for (int i = 0; i < n; i++) {
    int sh = rnd() % 64;
    a[i] = rnd() >> sh;
    b[i] = (uint32_t)(rnd() >> (32 + rnd() % 31)) | 1u;
    if (i % 3 == 0) b[i] = 1 + rnd() % 12;
    c[i] = (rnd() >> (rnd() % 60)) | 1ull;
    if (i % 5 == 0) c[i] = 262139;
  }
  uint64_t *da, *dc, *dq1, *dr1, *dq2, *dr2; uint32_t* db;
  hipMalloc(&da, n * 8); hipMalloc(&dc, n * 8); hipMalloc(&db, n * 4);
  hipMalloc(&dq1, n * 8); hipMalloc(&dr1, n * 8); hipMalloc(&dq2, n * 8); hipMalloc(&dr2, n * 8);
  hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(dc, c.data(), n * 8, hipMemcpyHostToDevice);
  hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(da, db, dc, dq1, dr1, dq2, dr2, n);
  hipMemcpy(q1.data(), dq1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(r1.data(), dr1, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(q2.data(), dq2, n * 8, hipMemcpyDeviceToHost); hipMemcpy(r2.data(), dr2, n * 8, hipMemcpyDeviceToHost);
  long bad = 0;
  for (int i = 0; i < n; i++) {
    if (q1[i] != a[i] / b[i] || r1[i] != a[i] % b[i] || q2[i] != a[i] / c[i] || r2[i] != a[i] % c[i]) {
      if (bad < 5) printf("MISMATCH a=%llu b=%u c=%llu dev q1=%llu r1=%llu q2=%llu r2=%llu\n", (unsigned long long)a[i], b[i], (unsigned long long)c[i],
                          (unsigned long long)q1[i], (unsigned long long)r1[i], (unsigned long long)q2[i], (unsigned long long)r2[i]);
      bad++;
    }
  }
  printf("divtest: %ld mismatches of %d\n", bad, n);
  return bad != 0;
}
