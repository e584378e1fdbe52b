/* Development tool (not part of the product): LD_PRELOAD this to learn WHY a process aborts.
 *
 *   gcc -O1 -g -shared -fPIC benchmarks/abort_bt.c -o benchmarks/abort_bt.so -ldl
 *   LD_PRELOAD=benchmarks/abort_bt.so python -m pytest -s -p no:faulthandler ...
 *
 * pytest captures fd 2, so "terminate called after throwing ..." and the HSA runtime's "Memory access fault by GPU ..."
 * lines never reach the terminal; Python's faulthandler shows Python frames only.  This library keeps a private duplicate
 * of the ORIGINAL stderr (taken in its constructor, before pytest redirects anything) and, on SIGABRT / SIGSEGV / SIGBUS,
 * writes the native backtrace of the faulting thread there.  Run pytest with -s as well, so that the runtime's own
 * last words (stdio inside libstdc++ / libhsa, not interposable) are not captured either.  */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

static int g_err = 2;

static void put(const char* s) { ssize_t r = write(g_err, s, strlen(s)); (void)r; }

static void handler(int sig, siginfo_t* info, void* uctx) {
    (void)uctx;
    char line[128];
    snprintf(line, sizeof line, "\n==== abort_bt: signal %d (code %d, addr %p) native backtrace of the faulting thread ====\n", sig,
             info ? info->si_code : 0, info ? info->si_addr : NULL);
    put(line);
    void* frames[96];
    int n = backtrace(frames, 96);
    backtrace_symbols_fd(frames, n, g_err);
    put("==== abort_bt: end ====\n");
    signal(sig, SIG_DFL);
    raise(sig);
}

__attribute__((constructor)) static void init(void) {
    g_err = dup(2);
    if (g_err < 0) g_err = 2;
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = handler;
    sa.sa_flags = SA_SIGINFO | SA_RESETHAND;
    sigaction(SIGABRT, &sa, NULL);
    sigaction(SIGSEGV, &sa, NULL);
    sigaction(SIGBUS, &sa, NULL);
    void* dummy[4];
    backtrace(dummy, 4); /* loads libgcc now: backtrace() must not dlopen inside a signal handler */
}
