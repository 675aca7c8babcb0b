/* LD_PRELOAD helper for tools/exit_probe.py: print the C stack (module + offset per frame) when the process
 * aborts or faults, so that an at-exit "double free" names the library whose destructor ran.  Diagnostics only. */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

static void on_fatal(int sig)
{
    void *bt[96];
    const char msg[] = "\n[abort_bt] fatal signal, C stack:\n";
    int n = backtrace(bt, 96);
    (void)!write(2, msg, sizeof(msg) - 1);
    backtrace_symbols_fd(bt, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}

__attribute__((constructor)) static void abort_bt_init(void)
{
    void *warm[4];
    backtrace(warm, 4);                 /* loads libgcc now, not inside the handler */
    signal(SIGABRT, on_fatal);
    signal(SIGSEGV, on_fatal);
}
