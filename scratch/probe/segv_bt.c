// LD_PRELOAD helper for the graph-destroy investigation: on SIGSEGV / SIGBUS / SIGABRT print the
// NATIVE backtrace (module + offset per frame, symbol where exported) of the faulting thread to stderr, then die with the
// default action.  Python's faulthandler only shows Python frames; rocgdb changes the timing.
//   gcc -shared -fPIC -O1 -o segv_bt.so segv_bt.c ; LD_PRELOAD=$PWD/segv_bt.so python -p no:faulthandler ...
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>

static void handler(int sig, siginfo_t* si, void* ctx) {
  (void)ctx;
  char head[160];
  int n = snprintf(head, sizeof head, "\n== segv_bt: signal %d, fault address %p, native backtrace of the faulting thread:\n", sig, si ? si->si_addr : 0);
  if (write(2, head, (size_t)n) < 0) {}
  void* bt[96];
  int d = backtrace(bt, 96);
  backtrace_symbols_fd(bt, d, 2);
  const char tail[] = "== segv_bt: end\n";
  if (write(2, tail, sizeof tail - 1) < 0) {}
  signal(sig, SIG_DFL);
  raise(sig);
}

__attribute__((constructor)) static void init(void) {
  struct sigaction sa;
  memset(&sa, 0, sizeof sa);
  sa.sa_sigaction = handler;
  sa.sa_flags = SA_SIGINFO | SA_RESETHAND;
  sigaction(SIGSEGV, &sa, 0);
  sigaction(SIGBUS, &sa, 0);
  sigaction(SIGABRT, &sa, 0);
  void* warm[4];
  backtrace(warm, 4);            // loads libgcc's unwinder now, not inside the handler
}
