#include <hip/hip_runtime.h>
__device__ __forceinline__ float shr1(float v) {   // lane i <- lane i-1 (lane 0 keeps 0)
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float shl1(float v) {   // lane i <- lane i+1
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, true));
}
__global__ void k(const float* in, float* out) {
  float v = in[threadIdx.x];
  out[threadIdx.x] = v + shr1(v) + shl1(v);
}
int main(){ float h[64], *d, *o; for(int i=0;i<64;i++)h[i]=i; hipMalloc(&d,256); hipMalloc(&o,256); hipMemcpy(d,h,256,hipMemcpyHostToDevice);
 hipLaunchKernelGGL(k,dim3(1),dim3(64),0,0,d,o); hipMemcpy(h,o,256,hipMemcpyDeviceToHost); for(int i=0;i<64;i++)printf("%g ",h[i]); printf("\n"); return 0; }
