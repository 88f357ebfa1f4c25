// One rank that is its own partner: a grouped ncclSend / ncclRecv inside a hipGraph capture, issued
//   mode "origin"          on the capturing stream itself,
//   mode "fork-nonblocking" on a second (hipStreamNonBlocking) stream forked into the capture with an event and
//                           joined back with another (what csrc/comm.hip did when RCCL 2.26 crashed, commit 820fc9e),
//   mode "fork-blocking"    the same with a default-flags second stream,
//   mode "fork-eager-first" fork-nonblocking, but after one eager (uncaptured) exchange on that stream.
// Prints a line per step; run under rocgdb to see where a crash comes from (tools/rccl_capture_probe.sh).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <string>

#define CK(x)                                                                                      \
  do                                                                                               \
  {                                                                                                \
    auto e_ = (x);                                                                                 \
    if ((int)e_ != 0)                                                                              \
    {                                                                                              \
      std::printf("FAILED %s -> %d (line %d)\n", #x, (int)e_, __LINE__);                           \
      std::fflush(stdout);                                                                         \
      return 2;                                                                                    \
    }                                                                                              \
  } while (0)
#define STEP(msg)                                                                                  \
  do                                                                                               \
  {                                                                                                \
    std::printf("  step: %s\n", msg);                                                              \
    std::fflush(stdout);                                                                           \
  } while (0)

int main(int argc, char** argv)
{
  const std::string mode = argc > 1 ? argv[1] : "origin";
  std::printf("mode %s\n", mode.c_str());
  CK(hipSetDevice(0));
  ncclUniqueId id;
  CK(ncclGetUniqueId(&id));
  ncclComm_t comm;
  CK(ncclCommInitRank(&comm, 1, id, 0));
  const size_t n = 1 << 16;
  double *a, *b;
  CK(hipMalloc(&a, n * sizeof(double)));
  CK(hipMalloc(&b, n * sizeof(double)));
  CK(hipMemset(a, 1, n * sizeof(double)));
  hipStream_t s, c;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&c, mode == "fork-blocking" ? hipStreamDefault : hipStreamNonBlocking));
  hipEvent_t e1, e2;
  CK(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
  auto exchange = [&](hipStream_t on) -> int {
    CK(ncclGroupStart());
    CK(ncclSend(a, n, ncclDouble, 0, comm, on));
    CK(ncclRecv(b, n, ncclDouble, 0, comm, on));
    CK(ncclGroupEnd());
    return 0;
  };
  if (mode == "fork-eager-first")
  {
    STEP("eager exchange on the second stream");
    if (exchange(c))
      return 2;
    CK(hipStreamSynchronize(c));
  }
  const bool fork = mode != "origin";
  STEP("begin capture");
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
  for (int rep = 0; rep < 2; ++rep) // two exchanges per capture, the events re-recorded, as in a V-cycle
  {
    if (fork)
    {
      STEP("fork: record on the capturing stream, second stream waits");
      CK(hipEventRecord(e1, s));
      CK(hipStreamWaitEvent(c, e1, 0));
    }
    STEP("grouped send/recv");
    if (exchange(fork ? c : s))
      return 2;
    if (fork)
    {
      STEP("join: record on the second stream, capturing stream waits");
      CK(hipEventRecord(e2, c));
      CK(hipStreamWaitEvent(s, e2, 0));
    }
  }
  STEP("end capture");
  hipGraph_t g = nullptr;
  CK(hipStreamEndCapture(s, &g));
  STEP("instantiate");
  hipGraphExec_t ge = nullptr;
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int r = 0; r < 3; ++r)
  {
    STEP("launch");
    CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
  }
  double h[2] = {0, 0}, ha[2] = {1, 1};
  CK(hipMemcpy(h, b, sizeof(h), hipMemcpyDeviceToHost));
  CK(hipMemcpy(ha, a, sizeof(ha), hipMemcpyDeviceToHost));
  std::printf("  received == sent: %s\n", std::memcmp(h, ha, sizeof(h)) == 0 ? "yes" : "NO");
  STEP("destroy");
  CK(hipGraphExecDestroy(ge));
  CK(hipGraphDestroy(g));
  CK(ncclCommDestroy(comm));
  std::printf("mode %s: OK\n", mode.c_str());
  return 0;
}
