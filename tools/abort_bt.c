/* abort_bt.c — LD_PRELOAD diagnostic (round 6): when glibc's heap check aborts a process ("free(): invalid next size", "corrupted size vs. prev_size"), print the
 * ABORTING thread's backtrace with the module each frame lives in, so that the free() that tripped can be attributed (rocprofiler-sdk, the HIP runtime, torch, this
 * library).  rocprofv3 installs its own SIGABRT handler after us; sigaction() is interposed so that ours stays first and theirs is chained.
 * build: gcc -shared -fPIC -O1 tools/abort_bt.c -o tools/libabort_bt.so -ldl      use: rocprofv3 --preload tools/libabort_bt.so ... (rocprofv3 builds the application's LD_PRELOAD itself) */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>
#include <sys/syscall.h>

static struct sigaction g_next;          /* whoever tried to take SIGABRT after us */
static int g_have_next = 0;
static int (*real_sigaction)(int, const struct sigaction*, struct sigaction*) = 0;

static void on_abort(int sig, siginfo_t* si, void* uc) {
  static volatile int once = 0;
  if (!__sync_lock_test_and_set(&once, 1)) {
    void* pc[48];
    char line[512];
    int n = backtrace(pc, 48);
    int len = snprintf(line, sizeof line, "\n[abort_bt] SIGABRT in tid %ld, %d frames:\n", (long)syscall(SYS_gettid), n);
    (void)!write(2, line, len);
    for (int i = 0; i < n; i++) {
      Dl_info di;
      memset(&di, 0, sizeof di);
      dladdr(pc[i], &di);
      const char* mod = di.dli_fname ? di.dli_fname : "?";
      const char* slash = strrchr(mod, '/');
      len = snprintf(line, sizeof line, "[abort_bt] #%02d %p %s+0x%lx %s\n", i, pc[i], slash ? slash + 1 : mod,
                     (unsigned long)((char*)pc[i] - (char*)di.dli_fbase), di.dli_sname ? di.dli_sname : "");
      (void)!write(2, line, len);
    }
  }
  if (g_have_next && (g_next.sa_flags & SA_SIGINFO) && g_next.sa_sigaction) { g_next.sa_sigaction(sig, si, uc); return; }
  if (g_have_next && g_next.sa_handler && g_next.sa_handler != SIG_DFL && g_next.sa_handler != SIG_IGN) { g_next.sa_handler(sig); return; }
  struct sigaction dfl;                  /* (not signal(): that is interposed below) */
  memset(&dfl, 0, sizeof dfl);
  dfl.sa_handler = SIG_DFL;
  real_sigaction(SIGABRT, &dfl, 0);
  raise(SIGABRT);
}

int sigaction(int sig, const struct sigaction* act, struct sigaction* old) {
  if (!real_sigaction) real_sigaction = (int (*)(int, const struct sigaction*, struct sigaction*))dlsym(RTLD_NEXT, "sigaction");
  if (sig == SIGABRT && act) {          /* keep ours installed, remember theirs and chain to it */
    if (old) { if (g_have_next) *old = g_next; else memset(old, 0, sizeof *old); }
    g_next = *act; g_have_next = 1;
    return 0;
  }
  return real_sigaction(sig, act, old);
}

typedef void (*handler_t)(int);
handler_t signal(int sig, handler_t h) {   /* the other way to take SIGABRT from us */
  if (sig == SIGABRT) {
    handler_t prev = g_have_next && !(g_next.sa_flags & SA_SIGINFO) ? g_next.sa_handler : SIG_DFL;
    memset(&g_next, 0, sizeof g_next); g_next.sa_handler = h; g_have_next = 1;
    return prev;
  }
  struct sigaction sa, old;
  memset(&sa, 0, sizeof sa);
  sa.sa_handler = h; sa.sa_flags = SA_RESTART; sigemptyset(&sa.sa_mask);
  if (!real_sigaction) real_sigaction = (int (*)(int, const struct sigaction*, struct sigaction*))dlsym(RTLD_NEXT, "sigaction");
  return real_sigaction(sig, &sa, &old) == 0 ? old.sa_handler : SIG_ERR;
}

__attribute__((constructor)) static void install(void) {
  void* warm[4];
  backtrace(warm, 4);                    /* loads libgcc now: not inside the handler */
  if (!real_sigaction) real_sigaction = (int (*)(int, const struct sigaction*, struct sigaction*))dlsym(RTLD_NEXT, "sigaction");
  struct sigaction sa;
  memset(&sa, 0, sizeof sa);
  sa.sa_sigaction = on_abort;
  sa.sa_flags = SA_SIGINFO;
  sigemptyset(&sa.sa_mask);
  real_sigaction(SIGABRT, &sa, 0);
}
