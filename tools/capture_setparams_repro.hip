// capture_setparams_repro.hip — HIP-only reproduction attempt of round 5's heap-check aborts under rocprofv3 ("What's weak" 3).
// What rounds 4-5 did while a stream was under capture: record one kernel node, then re-parameterise it once per further per-frame call
// (hipGraphKernelNodeSetParams with a DIFFERENT kernel function and a larger grid as the batch grew), moving the stream's capture dependencies in
// between (hipStreamUpdateCaptureDependencies).  This program does exactly that and nothing else — no library, no Python:
//   capture_setparams_repro edit   [graphs]   node editing under capture, 63 edits per graph        (the round-4/5 mechanism)
//   capture_setparams_repro deps   [graphs]   only the dependency updates, one node per call       (IndependentCall alone)
//   capture_setparams_repro plain  [graphs]   one node per batch, nothing edited                   (the round-6 mechanism)
//   capture_setparams_repro timed  [graphs]   20 kernel nodes + the event-record nodes vpp_graph_end(timed = 1) adds in front of the roots and behind the
//                                             leaves, hipEventElapsedTime after every replay         (what bench.py's timed regions are made of)
// Run each under `rocprofv3 --kernel-trace` a few times and count aborts (tools/capture_repro.sh).  Build: make -C tools capture_setparams_repro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); std::exit(2); } } while (0)

struct Frames { unsigned* p[64]; };
template <int VARIANT> __global__ __launch_bounds__(256) void batch_kernel(Frames f, int n, unsigned blocks_per_frame) {
  const unsigned frame = blockIdx.x / blocks_per_frame, b = blockIdx.x % blocks_per_frame;
  if ((int)frame < n) f.p[frame][b * 256 + threadIdx.x] += 1u + VARIANT * 0u;
}

int main(int argc, char** argv) {
  const char* mode = argc > 1 ? argv[1] : "edit";
  const int graphs = argc > 2 ? std::atoi(argv[2]) : 40;
  const bool edit = !std::strcmp(mode, "edit"), deps_only = !std::strcmp(mode, "deps"), timed = !std::strcmp(mode, "timed");
  const unsigned bpf = 64;
  Frames fr{};
  for (int k = 0; k < 64; k++) { CK(hipMalloc(&fr.p[k], bpf * 256 * sizeof(unsigned))); CK(hipMemset(fr.p[k], 0, bpf * 256 * sizeof(unsigned))); }
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  void* funcs[3] = {(void*)batch_kernel<0>, (void*)batch_kernel<1>, (void*)batch_kernel<2>};
  unsigned expect = 0;
  for (int g = 0; g < graphs; g++) {
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    hipGraphNode_t node = nullptr;
    if (edit) {
      int n = 1;
      batch_kernel<0><<<bpf * n, 256, 0, st>>>(fr, n, bpf);
      hipStreamCaptureStatus s; unsigned long long id; hipGraph_t gr; const hipGraphNode_t* deps; size_t nd;
      CK(hipStreamGetCaptureInfo_v2(st, &s, &id, &gr, &deps, &nd));
      node = deps[0];
      for (n = 2; n <= 64; n++) {   // one "per-frame call" each: the node carries one frame more, and another kernel instance from 2 and from 4 frames on
        Frames f2 = fr; int nn = n; unsigned b2 = bpf;
        void* args[3] = {&f2, &nn, &b2};
        hipKernelNodeParams kp{};
        kp.func = funcs[n >= 4 ? 2 : n >= 2 ? 1 : 0]; kp.gridDim = dim3(bpf * n); kp.blockDim = dim3(256); kp.kernelParams = args;
        CK(hipGraphKernelNodeSetParams(node, &kp));
        CK(hipStreamUpdateCaptureDependencies(st, &node, 1, hipStreamSetCaptureDependencies));   // (IndependentCall::absorbed_into -> rejoin)
      }
    } else if (deps_only) {
      std::vector<hipGraphNode_t> lane_last;
      for (int n = 1; n <= 64; n++) {   // every call its own node, two lanes side by side
        Frames f1{}; f1.p[0] = fr.p[n - 1];
        if (lane_last.size() == 2) CK(hipStreamUpdateCaptureDependencies(st, &lane_last[n & 1], 1, hipStreamSetCaptureDependencies));
        batch_kernel<0><<<bpf, 256, 0, st>>>(f1, 1, bpf);
        hipStreamCaptureStatus s; unsigned long long id; hipGraph_t gr; const hipGraphNode_t* deps; size_t nd;
        CK(hipStreamGetCaptureInfo_v2(st, &s, &id, &gr, &deps, &nd));
        if (lane_last.size() < 2) lane_last.push_back(deps[0]); else lane_last[n & 1] = deps[0];
        CK(hipStreamUpdateCaptureDependencies(st, lane_last.data(), lane_last.size(), hipStreamSetCaptureDependencies));
      }
    } else if (timed) {
      for (int k = 0; k < 20; k++) batch_kernel<2><<<bpf * 64, 256, 0, st>>>(fr, 64, bpf);
    } else {
      batch_kernel<2><<<bpf * 64, 256, 0, st>>>(fr, 64, bpf);
    }
    hipGraph_t graph; hipGraphExec_t exec;
    CK(hipStreamEndCapture(st, &graph));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (timed) {   // runtime.hip: vpp_graph_end(timed = 1)
      size_t nn = 0, ne = 0;
      CK(hipGraphGetNodes(graph, nullptr, &nn)); CK(hipGraphGetEdges(graph, nullptr, nullptr, &ne));
      std::vector<hipGraphNode_t> nodes(nn), from(ne), to(ne);
      CK(hipGraphGetNodes(graph, nodes.data(), &nn));
      if (ne) CK(hipGraphGetEdges(graph, from.data(), to.data(), &ne));
      CK(hipEventCreateWithFlags(&e0, hipEventReleaseToDevice)); CK(hipEventCreateWithFlags(&e1, hipEventReleaseToDevice));
      std::vector<hipGraphNode_t> roots, leaves;
      for (hipGraphNode_t n : nodes) {
        bool has_in = false, has_out = false;
        for (size_t k = 0; k < ne; k++) { has_in |= to[k] == n; has_out |= from[k] == n; }
        if (!has_in) roots.push_back(n);
        if (!has_out) leaves.push_back(n);
      }
      hipGraphNode_t n0 = nullptr, n1 = nullptr;
      CK(hipGraphAddEventRecordNode(&n0, graph, nullptr, 0, e0));
      for (hipGraphNode_t r : roots) CK(hipGraphAddDependencies(graph, &n0, &r, 1));
      CK(hipGraphAddEventRecordNode(&n1, graph, leaves.data(), leaves.size(), e1));
    }
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    for (int r = 0; r < 5; r++) {
      CK(hipGraphLaunch(exec, st));
      if (timed) { float ms = 0; CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); }
    }
    expect += timed ? 100 : 5;
    CK(hipStreamSynchronize(st));
    if (e0) { CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1)); }
    CK(hipGraphExecDestroy(exec));
    CK(hipGraphDestroy(graph));
  }
  std::vector<unsigned> h(bpf * 256);
  for (int k = 0; k < 64; k++) {
    CK(hipMemcpy(h.data(), fr.p[k], h.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
    for (unsigned v : h) if (v != expect) { std::fprintf(stderr, "frame %d: %u != %u\n", k, v, expect); return 1; }
  }
  std::printf("capture_setparams_repro %s: %d graphs ok\n", mode, graphs);
  return 0;
}
