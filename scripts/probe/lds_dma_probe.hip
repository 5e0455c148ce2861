// LDS-DMA as k_hash_select_hi uses it: two global_load_lds_dwordx4 per wave (256 + 16 words) into the wave's staging area,
// explicit s_waitcnt; checks the landed words.  hipcc --offload-arch=gfx950 -O3 -o lds_dma_probe lds_dma_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
__global__ void k(const uint32_t* src, uint32_t* dst)
{
  __shared__ __align__(16) uint32_t stage[4][272];
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t lds_off = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)(&stage[wv][0]);
  const uint32_t* base = src + 1000 * wv + 4; // (16-byte aligned, not more)
  const uint32_t* g0 = base + 4 * lane;
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g0), "s"(lds_off) : "m0", "memory");
  if (lane < 4) {
    const uint32_t* g1 = base + 256 + 4 * lane;
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g1), "s"(lds_off + 1024u) : "m0", "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (uint32_t i = lane; i < 272; i += 64) dst[wv * 272 + i] = stage[wv][i];
}
int main()
{
  std::vector<uint32_t> h(8192);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (uint32_t)(i * 2654435761u);
  uint32_t *s, *d;
  hipMalloc(&s, h.size() * 4);
  hipMalloc(&d, 4 * 272 * 4);
  hipMemcpy(s, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, s, d);
  std::vector<uint32_t> o(4 * 272);
  hipMemcpy(o.data(), d, o.size() * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int w = 0; w < 4; ++w)
    for (int i = 0; i < 272; ++i)
      if (o[w * 272 + i] != h[1000 * w + 4 + i]) ++bad;
  printf("bad=%d\n", bad);
  return bad != 0;
}
