/* segv_trace.c -- LD_PRELOAD helper for diagnosing a crash inside a runtime library: prints the NATIVE call stack on SIGSEGV
 * (glibc backtrace; function names where the library exports them) and re-raises.
 *   gcc -O1 -g -shared -fPIC tools/segv_trace.c -o tools/libsegv_trace.so
 *   LD_PRELOAD=tools/libsegv_trace.so python tools/cap_try.py 2 1 */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>

static void on_segv(int sig, siginfo_t* si, void* ctx)
{
    void* frames[64];
    (void)ctx;
    const char msg[] = "\n==== SIGSEGV: native backtrace ====\n";
    (void)!write(2, msg, sizeof msg - 1);
    char buf[64];
    int n = snprintf(buf, sizeof buf, "fault address %p\n", si ? si->si_addr : 0);
    (void)!write(2, buf, (size_t)n);
    n = backtrace(frames, 64);
    backtrace_symbols_fd(frames, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}

void segv_trace_install(void);
__attribute__((constructor)) static void install(void) { segv_trace_install(); }

/* also callable late (ctypes), after other libraries have installed their own handlers */
void segv_trace_install(void)
{
    /* an alternate stack: a stack OVERFLOW (runaway recursion) must still be able to run the handler */
    static char altstack[1 << 16];
    stack_t ss;
    ss.ss_sp = altstack; ss.ss_size = sizeof altstack; ss.ss_flags = 0;
    sigaltstack(&ss, 0);
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = on_segv;
    sa.sa_flags = SA_SIGINFO | SA_RESETHAND | SA_ONSTACK;
    sigaction(SIGSEGV, &sa, 0);
}
