// cap_repro.hip -- standalone reproducer of the stream-capture topology of yf_engine.hip's run_forward (no engine, no torch):
// empty kernels on the engine's exact stream / event pattern, captured from an origin stream and ended with hipStreamEndCapture.
//
//   origin --ev_fork--> side[0]                                   (two lanes: fork)
//   lane l (origin / side[0]) --ev_bfork[l]--> bside[l]           (the small head's branch: a fork nested in lane 1's fork)
//   bside[l] --ev_bjoin[l]--> lane l                              (branch joins its lane)
//   side[0] --ev_join--> origin                                   (lanes join the origin)
//
//   cap_repro LANES BRANCHES [fresh] [lanemajor] [blocking] [thread]
//     fresh      a new event for every record (instead of the engine's per-lane events, recorded once per pass)
//     lanemajor  issue lane 0's whole chain, then lane 1's (the engine issues op-major: op k of every lane, then op k + 1)
//     blocking   side streams created with hipStreamDefault instead of hipStreamNonBlocking
//     thread     hipStreamCaptureModeThreadLocal instead of Global (torch.cuda.graph's default is Global)
//     nopost     no launch on the lanes after the branch joins (yf_forward; yf_detect queues each chunk's decode + NMS there): lane 1's
//                join event is then recorded on a stream whose captured dependency set is TWO nodes (head_4 of the lane, head_5 of its
//                branch) with no node in between
//     joinorigin with nopost: the branches join the ORIGIN directly instead of their lane (every event is recorded behind one node)
//     postonbranch  yf_detect without a join INTO a forked lane: the lane joins its BRANCH stream (event), the chunk's decode + NMS runs
//                there, and branch and lane both join the origin
// Prints "ok nodes=<n>" and exits 0 when the capture ends, the graph instantiates and one replay completes.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#define CK(expr)                                                                               \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
            exit(3);                                                                           \
        }                                                                                      \
    } while (0)

__global__ void empty_kernel(int* p, int v)
{
    if (p && threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(p, v);
}

int main(int argc, char** argv)
{
    const int lanes = argc > 1 ? atoi(argv[1]) : 2, branches = argc > 2 ? atoi(argv[2]) : 1;
    bool fresh = false, lanemajor = false, blocking = false, threadmode = false, nopost = false, joinorigin = false, postonbranch = false;
    for (int i = 3; i < argc; ++i) {
        fresh |= !strcmp(argv[i], "fresh");
        lanemajor |= !strcmp(argv[i], "lanemajor");
        blocking |= !strcmp(argv[i], "blocking");
        threadmode |= !strcmp(argv[i], "thread");
        nopost |= !strcmp(argv[i], "nopost");
        joinorigin |= !strcmp(argv[i], "joinorigin");
        postonbranch |= !strcmp(argv[i], "postonbranch");
    }
    if (lanes < 1 || lanes > 4) return 2;
    const unsigned sflag = blocking ? hipStreamDefault : hipStreamNonBlocking;
    hipStream_t origin, side[3], bside[4];
    CK(hipStreamCreateWithFlags(&origin, hipStreamNonBlocking));   // torch's capture stream is a non-default pool stream
    for (auto& s : side) CK(hipStreamCreateWithFlags(&s, sflag));
    for (auto& s : bside) CK(hipStreamCreateWithFlags(&s, sflag));
    std::vector<hipEvent_t> pool;
    auto new_event = [&]() {
        hipEvent_t e;
        CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        pool.push_back(e);
        return e;
    };
    hipEvent_t ev_fork = new_event(), ev_join[3], ev_bfork[4], ev_bjoin[4], ev_l2b[4];
    for (auto& e : ev_l2b) e = new_event();
    for (auto& e : ev_join) e = new_event();
    for (auto& e : ev_bfork) e = new_event();
    for (auto& e : ev_bjoin) e = new_event();
    auto ev = [&](hipEvent_t fixed) { return fresh ? new_event() : fixed; };
    int* d_cnt;
    CK(hipMalloc(&d_cnt, 4));
    CK(hipMemset(d_cnt, 0, 4));

    // the engine's launch list: 17 ops on the lane, 2 branch ops (conv5_3+5_4, conv5_5+5_6+head_5), 4 more lane ops, then post
    const int n_pre = 17, n_branch = 2, n_post = 4;
    int launches = 0;
    auto issue = [&]() {
        if (lanes > 1) {
            hipEvent_t e = ev(ev_fork);
            CK(hipEventRecord(e, origin));
            for (int l = 1; l < lanes; ++l) CK(hipStreamWaitEvent(side[l - 1], e, 0));
        }
        bool forked[4] = {false, false, false, false};
        auto lane_stream = [&](int l) { return l == 0 ? origin : side[l - 1]; };
        auto op = [&](int k, int l) {
            hipStream_t s = lane_stream(l);
            const bool is_branch = k >= n_pre && k < n_pre + n_branch;
            if (branches && is_branch) {
                if (!forked[l]) {
                    hipEvent_t e = ev(ev_bfork[l]);
                    CK(hipEventRecord(e, s));
                    CK(hipStreamWaitEvent(bside[l], e, 0));
                    forked[l] = true;
                }
                s = bside[l];
            }
            hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s, d_cnt, 1);
            ++launches;
        };
        const int n_ops = n_pre + n_branch + n_post;
        if (lanemajor) {
            for (int l = 0; l < lanes; ++l)
                for (int k = 0; k < n_ops; ++k) op(k, l);
        } else {
            for (int k = 0; k < n_ops; ++k)
                for (int l = 0; l < lanes; ++l) op(k, l);
        }
        if (postonbranch) {
            for (int l = 0; l < lanes; ++l) {
                hipStream_t ps = lane_stream(l);
                if (forked[l]) {   // the lane joins its branch; the post runs there
                    hipEvent_t e = ev(ev_l2b[l]);
                    CK(hipEventRecord(e, lane_stream(l)));
                    CK(hipStreamWaitEvent(bside[l], e, 0));
                    ps = bside[l];
                }
                hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, ps, d_cnt, 1);
                ++launches;
                if (forked[l]) {
                    hipEvent_t e = ev(ev_bjoin[l]);
                    CK(hipEventRecord(e, bside[l]));
                    CK(hipStreamWaitEvent(origin, e, 0));
                }
            }
        }
        for (int l = 0; l < lanes && !postonbranch; ++l)
            if (forked[l]) {
                hipEvent_t e = ev(ev_bjoin[l]);
                CK(hipEventRecord(e, bside[l]));
                CK(hipStreamWaitEvent(joinorigin ? origin : lane_stream(l), e, 0));
            }
        for (int l = 0; l < lanes && !nopost && !postonbranch; ++l) {   // post-process of each chunk on its lane
            hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, lane_stream(l), d_cnt, 1);
            ++launches;
        }
        for (int l = 1; l < lanes; ++l) {
            hipEvent_t e = ev(ev_join[l - 1]);
            CK(hipEventRecord(e, side[l - 1]));
            CK(hipStreamWaitEvent(origin, e, 0));
        }
        CK(hipGetLastError());
    };

    issue();   // eager pass first (the engine is always run once before it is captured)
    CK(hipDeviceSynchronize());
    const int eager = launches;
    launches = 0;
    printf("capturing lanes=%d branches=%d fresh=%d lanemajor=%d blocking=%d thread=%d nopost=%d joinorigin=%d\n", lanes, branches, fresh, lanemajor,
           blocking, threadmode, nopost, joinorigin);
    fflush(stdout);
    CK(hipStreamBeginCapture(origin, threadmode ? hipStreamCaptureModeThreadLocal : hipStreamCaptureModeGlobal));
    issue();
    hipGraph_t graph = nullptr;
    printf("ending capture\n");
    fflush(stdout);
    CK(hipStreamEndCapture(origin, &graph));
    size_t nodes = 0;
    CK(hipGraphGetNodes(graph, nullptr, &nodes));
    hipGraphExec_t exec;
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    CK(hipGraphLaunch(exec, origin));
    CK(hipStreamSynchronize(origin));
    int cnt = 0;
    CK(hipMemcpy(&cnt, d_cnt, 4, hipMemcpyDeviceToHost));
    printf("ok nodes=%zu launches=%d counter=%d (expected %d)\n", nodes, launches, cnt, eager + launches);
    return cnt == eager + launches ? 0 : 4;
}
