// Reproducer for DESIGN.md section 5: "hipStreamEndCapture segfaults for fork/join patterns over >= 3 streams" (ROCm 7.2, MI355X).
// Issues the pattern schedule.run_schedule() issues under capture, in pure HIP: fork event e0 on the origin stream, side streams wait
// on it, a side stream waits on an event recorded by ANOTHER side stream, every side stream's tail is joined into the origin.
//   hipcc --offload-arch=gfx950 tools/repro_capture3.cpp -o /tmp/repro && /tmp/repro [n_streams] [cross]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k(float* p, float v) { p[threadIdx.x] += v; }
int main(int argc, char** argv) {
    const int S = argc > 1 ? atoi(argv[1]) : 3, cross = argc > 2 ? atoi(argv[2]) : 1;
    float* buf; CK(hipMalloc(&buf, 64 * 8 * sizeof(float))); CK(hipMemset(buf, 0, 64 * 8 * sizeof(float)));
    std::vector<hipStream_t> st(S); for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    std::vector<hipEvent_t> ev(4 * S); for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    CK(hipStreamBeginCapture(st[0], hipStreamCaptureModeThreadLocal));
    CK(hipEventRecord(ev[0], st[0]));                                   // fork
    for (int s = 1; s < S; ++s) CK(hipStreamWaitEvent(st[s], ev[0], 0));
    for (int s = 0; s < S; ++s) k<<<1, 64, 0, st[s]>>>(buf + 64 * s, 1.f);
    if (cross && S >= 3) {                                              // side stream 2 consumes what side stream 1 produced
        CK(hipEventRecord(ev[1], st[1]));
        CK(hipStreamWaitEvent(st[2], ev[1], 0));
        k<<<1, 64, 0, st[2]>>>(buf + 64, 2.f);
        k<<<1, 64, 0, st[1]>>>(buf + 64 * 3, 1.f);                       // stream 1 goes on after its record
    }
    for (int s = 1; s < S; ++s) {                                       // join every side stream's tail
        CK(hipEventRecord(ev[S + s], st[s]));
        CK(hipStreamWaitEvent(st[0], ev[S + s], 0));
    }
    k<<<1, 64, 0, st[0]>>>(buf, 1.f);
    hipGraph_t g; printf("ending capture (%d streams, cross=%d) ...\n", S, cross); fflush(stdout);
    CK(hipStreamEndCapture(st[0], &g));
    hipGraphExec_t ge; CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, st[0]));
    CK(hipStreamSynchronize(st[0]));
    float h[4]; for (int s = 0; s < 4 && s < S + 1; ++s) CK(hipMemcpy(&h[s], buf + 64 * s, sizeof(float), hipMemcpyDeviceToHost));
    printf("ok: 3 replays, buf[0]=%g buf[64]=%g buf[128]=%g\n", h[0], h[1], h[2]);
    return 0;
}
