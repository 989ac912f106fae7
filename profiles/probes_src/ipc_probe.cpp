// Can two processes on one GPU exchange a staging buffer with the copy engine?  Forks BEFORE any HIP call; the parent
// exports a device buffer (hipIpcGetMemHandle) and an interprocess event (hipIpcGetEventHandle), the child maps both,
// waits for the event on its stream and pulls the buffer with hipMemcpyAsync; 20 rounds with a host-side sequence number
// in a pipe (a wait must be enqueued after the record it is meant for).
// hipcc -O2 -o ipc_probe ipc_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <sys/wait.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("[%s] %s: %s\n", who, #x, hipGetErrorString(e_)); fflush(stdout); _exit(2); } } while (0)

int main()
{
  int p2c[2], c2p[2];
  if (pipe(p2c) || pipe(c2p)) return 1;
  const size_t n = 16u << 20;  // 64 MiB of floats
  pid_t pid = fork();
  const char* who = pid ? "parent" : "child";
  if (pid) {  // ---- parent: the producer
    float* buf;
    CK(hipMalloc(&buf, n * 4));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventInterprocess | hipEventDisableTiming));
    hipIpcMemHandle_t mh;
    hipIpcEventHandle_t eh;
    CK(hipIpcGetMemHandle(&mh, buf));
    CK(hipIpcGetEventHandle(&eh, ev));
    if (write(p2c[1], &mh, sizeof mh) != sizeof mh || write(p2c[1], &eh, sizeof eh) != sizeof eh) return 1;
    for (int k = 1; k <= 20; ++k) {
      int ack;
      if (k > 1 && read(c2p[0], &ack, 4) != 4) return 1;   // the child has pulled round k - 1 (host-level here: a probe)
      CK(hipMemsetD32Async((hipDeviceptr_t)buf, (int)k, n, s));
      CK(hipEventRecord(ev, s));
      if (write(p2c[1], &k, 4) != 4) return 1;             // "the record of round k has been enqueued"
    }
    int st;
    waitpid(pid, &st, 0);
    printf("parent: child exited with %d\n", WEXITSTATUS(st));
    return WEXITSTATUS(st);
  }
  // ---- child: the consumer
  hipIpcMemHandle_t mh;
  hipIpcEventHandle_t eh;
  if (read(p2c[0], &mh, sizeof mh) != sizeof mh || read(p2c[0], &eh, sizeof eh) != sizeof eh) return 1;
  float* peer;
  CK(hipIpcOpenMemHandle((void**)&peer, mh, hipIpcMemLazyEnablePeerAccess));
  hipEvent_t ev;
  CK(hipIpcOpenEventHandle(&ev, eh));
  float* mine;
  CK(hipMalloc(&mine, n * 4));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t t0, t1;
  CK(hipEventCreate(&t0));
  CK(hipEventCreate(&t1));
  std::vector<int> host(n);
  int bad = 0;
  for (int k = 1; k <= 20; ++k) {
    int seq;
    if (read(p2c[0], &seq, 4) != 4) return 1;
    CK(hipStreamWaitEvent(s, ev, 0));
    CK(hipEventRecord(t0, s));
    CK(hipMemcpyAsync(mine, peer, n * 4, hipMemcpyDeviceToDevice, s));
    CK(hipEventRecord(t1, s));
    CK(hipStreamSynchronize(s));
    float ms;
    CK(hipEventElapsedTime(&ms, t0, t1));
    CK(hipMemcpy(host.data(), mine, n * 4, hipMemcpyDeviceToHost));
    int wrong = 0;
    for (size_t i = 0; i < n; i += 4097) wrong += host[i] != seq;
    bad += wrong;
    if (k == 1 || k == 20 || wrong) printf("child: round %d pulled 64 MiB in %.3f ms (%.0f GB/s), %d wrong values\n", k, ms, n * 4 / ms / 1e6, wrong);
    if (write(c2p[1], &k, 4) != 4) return 1;
  }
  CK(hipIpcCloseMemHandle(peer));
  printf("child: %s\n", bad ? "FAILED" : "ok: interprocess event + copy-engine pull work");
  return bad ? 3 : 0;
}
