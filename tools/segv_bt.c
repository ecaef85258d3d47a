// tools/segv_bt.c -- debug aid: native backtrace on SIGSEGV / SIGABRT.  gcc -shared -fPIC -O1 -o tools/_build/segv_bt.so tools/segv_bt.c ;
// ORBX_SEGV_BT=$PWD/tools/_build/segv_bt.so python -m pytest tests -m gpu -s -p no:faulthandler   (tests/conftest.py installs it before every test:
// the HIP runtime replaces handlers installed earlier).  This is how the crash behind oracle/exports.map was found.
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <unistd.h>
static void handler(int sig, siginfo_t *si, void *ctx)
{
    void *bt[64];
    int n = backtrace(bt, 64);
    fprintf(stderr, "\n==== signal %d at address %p ====\n", sig, si->si_addr);
    backtrace_symbols_fd(bt, n, 2);
    _exit(139);
}
void segv_bt_install(void)
{
    struct sigaction sa;
    sa.sa_sigaction = handler;
    sigemptyset(&sa.sa_mask);
    sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
    sigaction(SIGSEGV, &sa, 0);
    sigaction(SIGABRT, &sa, 0);
}
