// Replays a dependency schedule (facenet_amd/schedule.py: steps = wait / run / record on a few streams) under HIP stream capture with
// trivial kernels: isolates the hipStreamEndCapture crash of DESIGN.md section 5 from PyTorch.
//   hipcc --offload-arch=gfx950 tools/repro_capture_steps.cpp -o build/repro_steps && build/repro_steps steps.txt [max_steps]
// steps.txt: first line "<n_streams> <n_events>", then one "<w|r|e> <stream> <index>" per line (w = wait event, r = run op, e = record event).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k(float* p) { p[threadIdx.x] += 1.f; }
int main(int argc, char** argv) {
    FILE* f = fopen(argv[1], "r"); if (!f) return 2;
    const long limit = argc > 2 ? atol(argv[2]) : 1L << 30;
    int S, E; if (fscanf(f, "%d %d", &S, &E) != 2) return 2;
    float* buf; CK(hipMalloc(&buf, 64 * sizeof(float) * S));
    std::vector<hipStream_t> st(S); for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    std::vector<hipEvent_t> ev(E + 1 + S); for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    std::vector<int> used(S, 0);
    CK(hipStreamBeginCapture(st[0], getenv("CAPTURE_GLOBAL") ? hipStreamCaptureModeGlobal : hipStreamCaptureModeThreadLocal));
    CK(hipEventRecord(ev[E], st[0]));
    for (int s = 1; s < S; ++s) CK(hipStreamWaitEvent(st[s], ev[E], 0));       // run_schedule: every side stream forks from e0
    char c; int s, x; long n = 0, runs = 0, waits = 0;
    while (n < limit && fscanf(f, " %c %d %d", &c, &s, &x) == 3) {
        if (c == 'w') { CK(hipStreamWaitEvent(st[s], ev[x], 0)); ++waits; }
        else if (c == 'e') CK(hipEventRecord(ev[x], st[s]));
        else { k<<<1, 64, 0, st[s]>>>(buf + 64 * s); ++runs; }
        ++n;
    }
    if (n >= limit) for (int t = 1; t < S; ++t) { CK(hipEventRecord(ev[E + t], st[t])); CK(hipStreamWaitEvent(st[0], ev[E + t], 0)); }   // truncated: join by hand
    hipGraph_t g; printf("%ld steps (%ld launches, %ld waits) on %d streams: ending capture ...\n", n, runs, waits, S); fflush(stdout);
    CK(hipStreamEndCapture(st[0], &g));
    size_t nn = 0; CK(hipGraphGetNodes(g, nullptr, &nn));
    hipGraphExec_t ge; CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st[0])); CK(hipStreamSynchronize(st[0]));
    printf("ok: %zu graph nodes, replayed\n", nn);
    return 0;
}
