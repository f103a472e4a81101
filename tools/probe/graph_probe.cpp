// probe: what do hipGraph*NodeGetParams return for nodes captured from hipMemcpyAsync / hipMemsetAsync / kernel launches?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void k(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.f; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s -> %s\n", #x, hipGetErrorString(e)); } } while (0)
int main() {
  float *a, *b; void* h;
  CK(hipMalloc(&a, 1 << 20)); CK(hipMalloc(&b, 1 << 20)); CK(hipHostMalloc(&h, 1 << 20, 0));
  hipStream_t s, s2; CK(hipStreamCreate(&s)); CK(hipStreamCreate(&s2));
  hipEvent_t e1, e2; CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
  CK(hipMemsetAsync(a, 0, 4096, s));
  CK(hipMemcpyAsync(b, a, 4096, hipMemcpyDeviceToDevice, s));
  CK(hipMemcpyAsync(a, h, 1000, hipMemcpyHostToDevice, s));
  CK(hipEventRecord(e1, s)); CK(hipStreamWaitEvent(s2, e1, 0));
  hipLaunchKernelGGL(k, dim3(4), dim3(256), 0, s2, b, 1000);
  CK(hipEventRecord(e2, s2)); CK(hipStreamWaitEvent(s, e2, 0));
  hipLaunchKernelGGL(k, dim3(4), dim3(256), 0, s, a, 1000);
  CK(hipMemcpyAsync(h, a, 512, hipMemcpyDeviceToHost, s));
  hipGraph_t g; CK(hipStreamEndCapture(s, &g));
  size_t n = 0; CK(hipGraphGetNodes(g, nullptr, &n));
  std::vector<hipGraphNode_t> nd(n); CK(hipGraphGetNodes(g, nd.data(), &n));
  size_t ne = 0; CK(hipGraphGetEdges(g, nullptr, nullptr, &ne));
  printf("nodes %zu edges %zu  a=%p b=%p h=%p\n", n, ne, a, b, h);
  for (size_t i = 0; i < n; ++i) {
    hipGraphNodeType t; CK(hipGraphNodeGetType(nd[i], &t));
    printf("node %zu type %d: ", i, (int)t);
    if (t == hipGraphNodeTypeMemcpy) {
      hipMemcpy3DParms p{}; hipError_t e = hipGraphMemcpyNodeGetParams(nd[i], &p);
      printf("memcpy rc=%d arrays %p %p src %p pitch %zu xs %zu ys %zu dst %p pitch %zu xs %zu ys %zu extent %zu %zu %zu kind %d srcPos %zu dstPos %zu\n", (int)e, (void*)p.srcArray, (void*)p.dstArray,
             p.srcPtr.ptr, p.srcPtr.pitch, p.srcPtr.xsize, p.srcPtr.ysize, p.dstPtr.ptr, p.dstPtr.pitch, p.dstPtr.xsize, p.dstPtr.ysize, p.extent.width, p.extent.height, p.extent.depth, (int)p.kind, p.srcPos.x, p.dstPos.x);
    } else if (t == hipGraphNodeTypeMemset) {
      hipMemsetParams p{}; hipError_t e = hipGraphMemsetNodeGetParams(nd[i], &p);
      printf("memset rc=%d dst %p elem %u w %zu h %zu pitch %zu val %u\n", (int)e, p.dst, p.elementSize, p.width, p.height, p.pitch, p.value);
    } else if (t == hipGraphNodeTypeKernel) {
      hipKernelNodeParams p{}; hipError_t e = hipGraphKernelNodeGetParams(nd[i], &p);
      hipFuncAttributes fa; hipError_t e2 = hipFuncGetAttributes(&fa, p.func);
      printf("kernel rc=%d func %p (host k=%p) attr rc=%d grid %u block %u args %p extra %p arg0 %p arg1 %d\n", (int)e, p.func, (void*)k, (int)e2, p.gridDim.x, p.blockDim.x, (void*)p.kernelParams,
             (void*)p.extra, p.kernelParams ? *(void**)p.kernelParams[0] : nullptr, p.kernelParams ? *(int*)p.kernelParams[1] : -1);
      // re-launch it eagerly from the extracted parameters
      hipError_t e3 = hipLaunchKernel(p.func, p.gridDim, p.blockDim, p.kernelParams, p.sharedMemBytes, s);
      printf("   relaunch rc=%d (%s)\n", (int)e3, hipGetErrorString(e3));
    } else printf("\n");
  }
  CK(hipStreamSynchronize(s));
  float v[4]; CK(hipMemcpy(v, b, 16, hipMemcpyDeviceToHost));
  printf("b[0]=%g (0 = never written, else relaunch count; uninitialised memory possible)\n", v[0]);
  return 0;
}
