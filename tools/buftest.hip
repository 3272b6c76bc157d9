// buftest.hip — what does gfx950 do with partially / scalar-offset out-of-range raw buffer accesses?  (measurement tooling)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(uint32_t* out, const uint8_t* s, uint8_t* d) {
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)s, 0, 64, 0x00020000);   // 64 bytes in range
  u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, 56, 0, 0);      // dwords at 56,60 in range; 64,68 out
  u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rs, 16, 64, 0);     // voffset in range, soffset moves it to 80..95
  u32x4 c = __builtin_amdgcn_raw_buffer_load_b128(rs, (uint32_t)-8, 0, 0);  // starts 8 bytes before the base
  u32x4 e = __builtin_amdgcn_raw_buffer_load_b128(rs, 64, 0, 0);      // wholly out
  u32x4 f = __builtin_amdgcn_raw_buffer_load_b128(rs, 48, 0, 0);      // wholly in
  uint32_t* o = out;
  for (int i = 0; i < 4; i++) { o[i] = a[i]; o[4 + i] = b[i]; o[8 + i] = c[i]; o[12 + i] = e[i]; o[16 + i] = f[i]; }
  __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)d, 0, 64, 0x00020000);
  u32x4 v = {0x11111111u, 0x22222222u, 0x33333333u, 0x44444444u};
  __builtin_amdgcn_raw_buffer_store_b128(v, rd, 56, 0, 0);    // partially out: which dwords land?
  __builtin_amdgcn_raw_buffer_store_b128(v, rd, 16, 128, 0);  // soffset beyond num_records
}
int main() {
  uint8_t h[512]; for (int i = 0; i < 512; i++) h[i] = (uint8_t)i;
  uint8_t *s, *d; uint32_t* o; hipMalloc(&s, 4096); hipMalloc(&d, 4096); hipMalloc(&o, 256);
  hipMemcpy(s + 256, h, 512, hipMemcpyHostToDevice); hipMemset(d, 0, 4096);
  k<<<1, 1>>>(o, s + 256 + 16, d);   // base = byte 16 of the pattern
  uint32_t r[20]; hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
  const char* nm[5] = {"partial (56..71 of 64)", "voff 16 + soff 64", "voff -8", "wholly out", "wholly in"};
  for (int t = 0; t < 5; t++) printf("%-24s %08x %08x %08x %08x\n", nm[t], r[4 * t], r[4 * t + 1], r[4 * t + 2], r[4 * t + 3]);
  uint8_t hd[256]; hipMemcpy(hd, d, 256, hipMemcpyDeviceToHost);
  printf("store partial: d[56..71] ="); for (int i = 56; i < 72; i++) printf(" %02x", hd[i]); printf("\nstore soff 128: d[144..159] ="); for (int i = 144; i < 160; i++) printf(" %02x", hd[i]); printf("\n");
  return 0;
}
