/* LD_PRELOAD helper (diagnostic): a C-level backtrace on SIGSEGV, for crashes inside libhipnlp.so / the HIP runtime under pytest.
 *   gcc -shared -fPIC -O1 -o tools/diag/_build/libsegv_bt.so tools/diag/segv_bt.c ; LD_PRELOAD=... python -m pytest ... */
#define _GNU_SOURCE
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>
static void handler(int sig, siginfo_t* si, void* ctx) {
    (void)ctx;
    void* frames[64];
    char msg[128];
    int n = snprintf(msg, sizeof msg, "\n== signal %d at address %p: C backtrace ==\n", sig, si->si_addr);
    if (write(2, msg, (size_t)n) < 0) {}
    n = backtrace(frames, 64);
    backtrace_symbols_fd(frames, n, 2);
    {   /* the mapping the faulting address lies in (or its neighbours): /proc/self/maps, raw */
        int fd = open("/proc/self/maps", O_RDONLY);
        if (fd >= 0) {
            static char buf[1 << 16];
            ssize_t r;
            if (write(2, "== /proc/self/maps ==\n", 22) < 0) {}
            while ((r = read(fd, buf, sizeof buf)) > 0) if (write(2, buf, (size_t)r) < 0) break;
            close(fd);
        }
    }
    signal(sig, SIG_DFL);
    raise(sig);
}
__attribute__((constructor)) static void install(void) {
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = handler;
    sa.sa_flags = SA_SIGINFO | SA_RESETHAND;
    sigaction(SIGSEGV, &sa, NULL);
    sigaction(SIGABRT, &sa, NULL);   /* glibc's heap checks (MALLOC_CHECK_=3) end in abort() */
    sigaction(SIGBUS, &sa, NULL);
}
