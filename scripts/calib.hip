// calibration microbenchmarks (dev tool): clocks, launch gap, memory latency on the target box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
__global__ void spin(long n, long* out){ long c0=clock64(), w0=wall_clock64(); float a=1.f; for(long i=0;i<n;i++) a=a*1.0000001f+1e-9f; long c1=clock64(), w1=wall_clock64(); if(threadIdx.x==0){out[0]=c1-c0; out[1]=w1-w0; out[2]=(long)a;} }
__global__ void empty(){}
__global__ void chase(const int* p, int n, long* out){ int i=0; long c0=wall_clock64(); for(int k=0;k<n;k++) i=p[i]; long c1=wall_clock64(); out[0]=c1-c0; out[1]=i; }
__global__ void small_row(const float* x, float* y, int C){ __shared__ float red[4]; float s=0; for(int c=threadIdx.x;c<C;c+=256) s+=x[c]; for(int m=32;m>=1;m>>=1) s+=__shfl_xor(s,m,64); __syncthreads(); if((threadIdx.x&63)==0) red[threadIdx.x>>6]=s; __syncthreads(); float t=red[0]+red[1]+red[2]+red[3]; for(int c=threadIdx.x;c<C;c+=256) y[c]=x[c]*t; }
int main(){
  long* d; CK(hipMalloc(&d, 64)); long h[4];
  hipStream_t st; CK(hipStreamCreate(&st));
  for(int rep=0;rep<3;rep++){ hipLaunchKernelGGL(spin,dim3(1),dim3(64),0,st,20000000L,d); CK(hipStreamSynchronize(st)); CK(hipMemcpy(h,d,32,hipMemcpyDeviceToHost)); printf("spin 1 wave: clock64=%ld wall(100MHz)=%ld -> sclk ~ %.0f MHz, cyc/iter %.2f\n", h[0],h[1], (double)h[0]/h[1]*100.0, (double)h[0]/20000000.0);}  
  hipLaunchKernelGGL(spin,dim3(1024),dim3(256),0,st,5000000L,d); CK(hipStreamSynchronize(st)); CK(hipMemcpy(h,d,32,hipMemcpyDeviceToHost)); printf("spin full chip: sclk ~ %.0f MHz\n",(double)h[0]/h[1]*100.0);
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  // launch gap: graph of 1000 empty kernels
  hipGraph_t g; hipGraphExec_t ge; CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal)); for(int i=0;i<1000;i++) hipLaunchKernelGGL(empty,dim3(1),dim3(64),0,st); CK(hipStreamEndCapture(st,&g)); CK(hipGraphInstantiate(&ge,g,nullptr,nullptr,0));
  CK(hipGraphLaunch(ge,st)); CK(hipStreamSynchronize(st));
  CK(hipEventRecord(e0,st)); CK(hipGraphLaunch(ge,st)); CK(hipEventRecord(e1,st)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms,e0,e1)); printf("graph: empty kernel chain: %.2f us per kernel\n", ms);
  // small_row chain
  float *x,*y; CK(hipMalloc(&x,1<<20)); CK(hipMalloc(&y,1<<20)); CK(hipMemset(x,0,1<<20));
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal)); for(int i=0;i<1000;i++) hipLaunchKernelGGL(small_row,dim3(1),dim3(256),0,st,x,y,2560); CK(hipStreamEndCapture(st,&g)); CK(hipGraphInstantiate(&ge,g,nullptr,nullptr,0));
  CK(hipGraphLaunch(ge,st)); CK(hipStreamSynchronize(st));
  CK(hipEventRecord(e0,st)); CK(hipGraphLaunch(ge,st)); CK(hipEventRecord(e1,st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms,e0,e1)); printf("graph: small_row(1 block, C=2560) chain: %.2f us per kernel\n", ms);
  // pointer chase
  for (size_t bytes : {(size_t)1<<20, (size_t)16<<20, (size_t)1<<30}) { int n=bytes/4; std::vector<int> hp(n); size_t stride=64*1031; // lines
    for(int i=0;i<n;i++) hp[i]=0; size_t cur=0; int steps=20000; for(int k=0;k<steps;k++){ size_t nx=(cur+stride)%n; nx-=nx%32; hp[cur]=(int)nx; cur=nx; }
    int* dp; CK(hipMalloc(&dp,bytes)); CK(hipMemcpy(dp,hp.data(),bytes,hipMemcpyHostToDevice));
    hipLaunchKernelGGL(chase,dim3(1),dim3(1),0,st,dp,steps,d); CK(hipStreamSynchronize(st)); hipLaunchKernelGGL(chase,dim3(1),dim3(1),0,st,dp,steps,d); CK(hipStreamSynchronize(st)); CK(hipMemcpy(h,d,16,hipMemcpyDeviceToHost)); printf("chase %zu MB: %.1f ns per load\n", bytes>>20, h[0]*10.0/steps); CK(hipFree(dp)); }
  return 0; }
