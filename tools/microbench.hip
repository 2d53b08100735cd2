// microbench.hip -- measures the primitives the graph path is built from on MI355X:
// scattered device-scope atomics, scattered sector loads/stores, LDS atomics.  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <chrono>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)

__device__ __forceinline__ uint64_t mix(uint64_t x){ x^=x>>33; x*=0xff51afd7ed558ccdULL; x^=x>>33; x*=0xc4ceb9fe1a85ec53ULL; x^=x>>33; return x; }

template<int MODE>
__global__ __launch_bounds__(256) void k_scatter(unsigned long long* a, uint64_t mask, uint64_t n, unsigned long long* sink){
  uint64_t i=(uint64_t)blockIdx.x*blockDim.x+threadIdx.x, stride=(uint64_t)gridDim.x*blockDim.x;
  unsigned long long acc=0;
  for(;i<n;i+=stride){
    uint64_t j=mix(i)&mask;
    if(MODE==0) atomicAdd(&a[j],1ull);                       // no-return 64-bit add
    else if(MODE==1) acc+=atomicAdd(&a[j],1ull);             // returning add
    else if(MODE==2) atomicMin(&a[j],(unsigned long long)i);  // no-return min
    else if(MODE==3) acc+=a[j];                              // 8-byte load
    else if(MODE==4) { ulonglong2 v=*reinterpret_cast<ulonglong2*>(&a[j&~3ull]); ulonglong2 w=*reinterpret_cast<ulonglong2*>(&a[(j&~3ull)+2]); acc+=v.x+v.y+w.x+w.y; } // 32-byte sector load
    else if(MODE==5) a[j]=i;                                 // 8-byte store
    else if(MODE==6) { unsigned int* p=(unsigned int*)a; atomicOr(&p[mix(i)&(mask*2+1)],1u<<(i&31)); } // 32-bit or
    else if(MODE==7) { ulonglong2 v=*reinterpret_cast<ulonglong2*>(&a[j&~3ull]); acc+=v.x; atomicAdd(&a[(j&~3ull)+3],1ull);} // load + add same sector (the sweep's pattern)
    else if(MODE==8) atomicCAS(&a[j],~0ull,(unsigned long long)i);
  }
  if(acc==0x1234567) *sink=acc;
}

__global__ __launch_bounds__(256) void k_lds(uint64_t n, int iters, unsigned long long* sink){
  __shared__ unsigned long long t[4096];
  for(int j=threadIdx.x;j<4096;j+=256) t[j]=0;
  __syncthreads();
  uint64_t i=(uint64_t)blockIdx.x*blockDim.x+threadIdx.x;
  for(int it=0;it<iters;++it){ uint64_t j=mix(i+it*0x9E3779B97F4A7C15ULL)&4095; atomicAdd(&t[j],1ull); }
  __syncthreads();
  if(t[threadIdx.x]==0x1234567) *sink=1;
}

int main(){
  const char* names[]={"atomicAdd64 noret","atomicAdd64 ret","atomicMin64 noret","load 8B","load 32B sector","store 8B","atomicOr32 noret","load32B+atomicAdd same sector","atomicCAS64"};
  unsigned long long* sink; CK(hipMalloc(&sink,8));
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for(uint64_t words : {1ull<<19, 1ull<<24, 1ull<<28, 1ull<<31, 1ull<<33}){   // 4 MiB, 128 MiB, 2 GiB, 16 GiB, 64 GiB
    unsigned long long* a; if(hipMalloc(&a,words*8)!=hipSuccess){printf("alloc %llu failed\n",(unsigned long long)words);continue;}
    CK(hipMemset(a,0,words*8));
    uint64_t n=1ull<<30;
    for(int mode=0;mode<9;++mode){
      float best=1e30f;
      for(int rep=0;rep<2;++rep){
        hipEventRecord(e0);
        switch(mode){
          case 0: k_scatter<0><<<2048,256>>>(a,words-1,n,sink);break; case 1: k_scatter<1><<<2048,256>>>(a,words-1,n,sink);break;
          case 2: k_scatter<2><<<2048,256>>>(a,words-1,n,sink);break; case 3: k_scatter<3><<<2048,256>>>(a,words-1,n,sink);break;
          case 4: k_scatter<4><<<2048,256>>>(a,words-1,n,sink);break; case 5: k_scatter<5><<<2048,256>>>(a,words-1,n,sink);break;
          case 6: k_scatter<6><<<2048,256>>>(a,words-1,n,sink);break; case 7: k_scatter<7><<<2048,256>>>(a,words-1,n,sink);break;
          case 8: k_scatter<8><<<2048,256>>>(a,words-1,n,sink);break; }
        hipEventRecord(e1); CK(hipEventSynchronize(e1)); float ms; hipEventElapsedTime(&ms,e0,e1); if(ms<best)best=ms;
      }
      printf("footprint %6.0f MiB  %-32s %7.2f ms  %6.2f G ops/s\n", words*8/1048576.0, names[mode], best, n/best/1e6);
    }
    hipFree(a);
  }
  { int iters=256; uint64_t nthreads=2048ull*256; float ms;
    hipEventRecord(e0); k_lds<<<2048,256>>>(nthreads,iters,sink); hipEventRecord(e1); CK(hipEventSynchronize(e1)); hipEventElapsedTime(&ms,e0,e1);
    hipEventRecord(e0); k_lds<<<2048,256>>>(nthreads,iters,sink); hipEventRecord(e1); CK(hipEventSynchronize(e1)); hipEventElapsedTime(&ms,e0,e1);
    printf("LDS atomicAdd64 random 4096 words/WG: %.2f ms  %.1f G ops/s\n", ms, nthreads*iters/ms/1e6); }
  // allocation cost
  for(uint64_t gib : {1ull,16ull,64ull}){
    void* p; hipEvent_t a0; (void)a0;
    auto t0=std::chrono::steady_clock::now(); hipError_t e=hipMalloc(&p,gib<<30); auto t1=std::chrono::steady_clock::now();
    if(e!=hipSuccess){printf("malloc %llu GiB failed\n",(unsigned long long)gib);continue;}
    hipMemset(p,0xFF,gib<<30); hipDeviceSynchronize(); auto t2=std::chrono::steady_clock::now();
    hipFree(p); auto t3=std::chrono::steady_clock::now();
    printf("hipMalloc %llu GiB: %.1f ms, memset %.1f ms, hipFree %.1f ms\n",(unsigned long long)gib,
      std::chrono::duration<double,std::milli>(t1-t0).count(), std::chrono::duration<double,std::milli>(t2-t1).count(), std::chrono::duration<double,std::milli>(t3-t2).count());
  }
  return 0;
}
