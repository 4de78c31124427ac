// Check of the two mechanisms the generated multiplication block of the final exponentiation relies on (gen_step_asm.py): a private-segment
// address handed to inline assembly as an SGPR (readfirstlane of the addrspace(5) pointer) and read with scratch_load, and a subroutine inside
// one asm block (s_call_b64 / s_setpc_b64 with block-local labels).  Prints "ok" or the first mismatch.  Not part of the library.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(int* out, int n) {
  int priv[64];
  for (int i = 0; i < 64; ++i) priv[i] = i * n + (int)threadIdx.x;
  int idx = n & 7;
  const uint32_t pa = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(5))) void*)&priv[idx * 4]);
  int r;
  asm volatile(
    "scratch_load_dwordx4 v[10:13], off, %1 offset:4\n"
    "s_waitcnt vmcnt(0)\n"
    "s_call_b64 s[26:27], Lsub_%=\n"
    "s_branch Lend_%=\n"
    "Lsub_%=:\n"
    "v_add_u32 v10, v10, v11\n"
    "s_setpc_b64 s[26:27]\n"
    "Lend_%=:\n"
    "v_mov_b32 %0, v10\n"
    : "=v"(r) : "s"(pa) : "v10", "v11", "v12", "v13", "s26", "s27", "memory");
  out[threadIdx.x + blockIdx.x * blockDim.x] = r;
}
int main() {
  int* d; hipMalloc(&d, 4 * 512);
  for (int n = 3; n < 12; ++n) {
    k<<<2, 256>>>(d, n);
    int h[512]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int idx = n & 7;
    for (int t = 0; t < 512; ++t) {
      int want = ((idx * 4 + 1) * n + (t & 255)) + ((idx * 4 + 2) * n + (t & 255));
      if (h[t] != want) { printf("MISMATCH n=%d t=%d got %d want %d\n", n, t, h[t], want); return 1; }
    }
  }
  printf("ok\n");
  return 0;
}
