/* DEV TOOL: LD_PRELOAD helper — print a native backtrace when the process aborts (a C++ terminate, a runtime assertion or a
 * glibc heap check inside a ctypes call leaves only Python frames in faulthandler's report).  Interposes abort(), keeps its
 * own SIGABRT handler installed (later sigaction / signal calls for SIGABRT are recorded and chained, not honoured).
 *   gcc -shared -fPIC -o /tmp/abort_bt.so tools/probe/abort_bt.c -ldl && LD_PRELOAD=/tmp/abort_bt.so python -m pytest ... */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <execinfo.h>
#include <signal.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
static struct sigaction chained;
static int have_chained = 0;
#include <fcntl.h>
static void dump(const char* why) {  // pytest redirects fd 2 into its capture file: also append to $ABORT_BT_LOG
  void* frames[64];
  int n = backtrace(frames, 64);
  const char* path = getenv("ABORT_BT_LOG");
  int fds[2] = {2, path ? open(path, O_WRONLY | O_CREAT | O_APPEND, 0644) : -1};
  for (int k = 0; k < 2; k++) {
    if (fds[k] < 0) continue;
    write(fds[k], why, strlen(why));
    backtrace_symbols_fd(frames, n, fds[k]);
  }
}
static void on_sig(int sig, siginfo_t* info, void* ctx) {
  dump("\n[abort_bt] SIGABRT handler, native backtrace:\n");
  if (have_chained && (chained.sa_flags & SA_SIGINFO) && chained.sa_sigaction) chained.sa_sigaction(sig, info, ctx);
  else if (have_chained && chained.sa_handler && chained.sa_handler != SIG_DFL && chained.sa_handler != SIG_IGN) chained.sa_handler(sig);
  signal(sig, SIG_DFL);
  raise(sig);
}
typedef int (*sigaction_fn)(int, const struct sigaction*, struct sigaction*);
static sigaction_fn real_sigaction(void) {
  static sigaction_fn f;
  if (!f) f = (sigaction_fn)dlsym(RTLD_NEXT, "sigaction");
  return f;
}
int sigaction(int sig, const struct sigaction* act, struct sigaction* old) {
  if (sig == SIGABRT && act) {
    if (old) *old = chained;
    chained = *act;
    have_chained = 1;
    return 0;
  }
  return real_sigaction()(sig, act, old);
}
void abort(void) {
  dump("\n[abort_bt] abort() called, native backtrace:\n");
  struct sigaction dfl;
  memset(&dfl, 0, sizeof(dfl));
  dfl.sa_handler = SIG_DFL;
  real_sigaction()(SIGABRT, &dfl, NULL);
  raise(SIGABRT);
  _exit(134);
}
__attribute__((constructor)) static void init(void) {
  struct sigaction sa;
  memset(&sa, 0, sizeof(sa));
  sa.sa_sigaction = on_sig;
  sa.sa_flags = SA_SIGINFO;
  real_sigaction()(SIGABRT, &sa, NULL);
}
