// lds_bench.hip -- LDS atomic rates on MI355X by operand width and operation (random addresses in a 128 KB array,
// one workgroup of 1024 threads per CU).  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint64_t mix(uint64_t x){ x^=x>>33; x*=0xff51afd7ed558ccdULL; x^=x>>33; x*=0xc4ceb9fe1a85ec53ULL; x^=x>>33; return x; }
template<int MODE>
__global__ __launch_bounds__(1024) void k(int iters, unsigned long long* sink){
  __shared__ unsigned long long t[16384];
  for(int j=threadIdx.x;j<16384;j+=1024) t[j]=~0ull;
  __syncthreads();
  uint32_t* t32=(uint32_t*)t;
  uint64_t i=(uint64_t)blockIdx.x*blockDim.x+threadIdx.x;
  unsigned long long acc=0;
  for(int it=0;it<iters;++it){
    uint64_t r=mix(i+it*0x9E3779B97F4A7C15ULL);
    uint32_t j=(uint32_t)r&16383u;
    if(MODE==0) atomicMin(&t[j],(unsigned long long)(r>>14));        // 64-bit min, no return
    else if(MODE==1) atomicMin(&t32[j*2],(uint32_t)(r>>14));         // 32-bit min, no return
    else if(MODE==2) acc+=atomicAdd(&t32[j*2],1u);                   // 32-bit add, returning
    else if(MODE==3) atomicAdd(&t32[j*2],1u);                        // 32-bit add
    else if(MODE==4) acc+=atomicCAS(&t[j],~0ull,(unsigned long long)(r>>14)); // 64-bit cas returning
    else if(MODE==5) acc+=t[j];                                      // 64-bit load
    else if(MODE==6) t[j]=r;                                         // 64-bit store
    else if(MODE==7) atomicAdd(&t[j],1ull);                          // 64-bit add
    else if(MODE==8) atomicOr(&t32[(uint32_t)r&32767u],1u<<(r>>20&31));  // 32-bit or
  }
  __syncthreads();
  if(acc==0x1234567||t[threadIdx.x]==0x1234567) *sink=acc;
}
int main(){
  const char* names[]={"min64","min32","add32 ret","add32","cas64 ret","load64","store64","add64","or32"};
  unsigned long long* sink; hipMalloc(&sink,8);
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int iters=4096; double n=256.0*1024*iters;
  for(int mode=0;mode<9;++mode){ float best=1e30f;
    for(int rep=0;rep<3;++rep){ hipEventRecord(e0);
      switch(mode){case 0:k<0><<<256,1024>>>(iters,sink);break;case 1:k<1><<<256,1024>>>(iters,sink);break;case 2:k<2><<<256,1024>>>(iters,sink);break;
        case 3:k<3><<<256,1024>>>(iters,sink);break;case 4:k<4><<<256,1024>>>(iters,sink);break;case 5:k<5><<<256,1024>>>(iters,sink);break;
        case 6:k<6><<<256,1024>>>(iters,sink);break;case 7:k<7><<<256,1024>>>(iters,sink);break;case 8:k<8><<<256,1024>>>(iters,sink);break;}
      hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms,e0,e1); if(ms<best)best=ms; }
    printf("%-10s %8.3f ms  %8.1f G lane-ops/s  %.2f cycles per wave-instruction per CU (2.4 GHz)\n",names[mode],best,n/best/1e6, best*1e-3*2.4e9/(16.0*iters)); }
  return 0; }
