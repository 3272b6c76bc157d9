// Issue rate of v_qsad_pk_u16_u8 (4 sliding 4-byte SADs per instruction, packed u16 accumulators) against v_sad_u8 on gfx950, and a check of
// its semantics against a byte loop.  hipcc --offload-arch=gfx950 -O3 qsad_rate.hip -o qsad_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
template <int MODE>
__global__ __launch_bounds__(256) void rate(uint64_t* out, uint32_t seed, int iters) {
  uint64_t s0 = seed * 0x9E3779B97F4A7C15ull + threadIdx.x, acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
  uint32_t s1 = seed ^ threadIdx.x, a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  for (int i = 0; i < iters; i++) {
    if (MODE == 0) {
      acc0 = __builtin_amdgcn_qsad_pk_u16_u8(s0, s1, acc0); acc1 = __builtin_amdgcn_qsad_pk_u16_u8(s0 + 1, s1, acc1);
      acc2 = __builtin_amdgcn_qsad_pk_u16_u8(s0 + 2, s1, acc2); acc3 = __builtin_amdgcn_qsad_pk_u16_u8(s0 + 3, s1, acc3);
    } else {
      a0 = __builtin_amdgcn_sad_u8((uint32_t)s0, s1, a0); a1 = __builtin_amdgcn_sad_u8((uint32_t)s0 + 1, s1, a1);
      a2 = __builtin_amdgcn_sad_u8((uint32_t)s0 + 2, s1, a2); a3 = __builtin_amdgcn_sad_u8((uint32_t)s0 + 3, s1, a3);
    }
    s0 += 0x0101010101010101ull;
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc0 + acc1 + acc2 + acc3 + a0 + a1 + a2 + a3;
}
__global__ void sem(uint64_t* out, uint64_t s0, uint32_t s1, uint64_t s2) { out[0] = __builtin_amdgcn_qsad_pk_u16_u8(s0, s1, s2); }
int main() {
  uint64_t* d; hipMalloc(&d, 1024 * 256 * 8);
  const uint64_t s0 = 0x80FF10203A004511ull; const uint32_t s1 = 0x00FE7F01u; const uint64_t s2 = 0x0001000200030004ull;
  sem<<<1, 1>>>(d, s0, s1, s2); uint64_t got; hipMemcpy(&got, d, 8, hipMemcpyDeviceToHost);
  uint64_t want = 0;
  for (int i = 0; i < 4; i++) { unsigned sum = (unsigned)((s2 >> (16 * i)) & 0xFFFF); for (int j = 0; j < 4; j++) { int a = (int)((s0 >> (8 * (i + j))) & 0xFF), b = (int)((s1 >> (8 * j)) & 0xFF); sum += (unsigned)(a > b ? a - b : b - a); } want |= (uint64_t)(sum & 0xFFFF) << (16 * i); }
  std::printf("qsad_pk_u16_u8 semantics: got %016llx, byte loop %016llx %s\n", (unsigned long long)got, (unsigned long long)want, got == want ? "(equal: unmasked, zero bytes count)" : "(DIFFERENT)");
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4096;
  for (int mode = 0; mode < 2; mode++) {
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
      hipEventRecord(e0);
      if (mode == 0) rate<0><<<1024, 256>>>(d, rep, iters); else rate<1><<<1024, 256>>>(d, rep, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const double inst = 1024.0 * 4 /*waves per block*/ * iters * 4;
    std::printf("%s: %.3f ms for %.0f wave-instructions -> %.2f cycles per instruction per SIMD at 2.4 GHz (1024 SIMDs)\n", mode == 0 ? "v_qsad_pk_u16_u8" : "v_sad_u8", best, inst,
                best * 1e-3 * 2.4e9 / (inst / 1024.0));
  }
  return 0;
}
