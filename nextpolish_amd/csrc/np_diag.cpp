// Abort diagnostics, off unless NP_ABORT_TRACE is set (tests/conftest.py sets it).
//
// A process that dies of SIGABRT inside native code says nothing about where: Python's faulthandler prints the Python frames, glibc and
// libstdc++ print one line, the HIP runtime sometimes none.  With NP_ABORT_TRACE set, loading either library installs
//   * a std::terminate handler that prints the type and what() of the exception in flight, and
//   * a SIGABRT handler that prints the native frames of the aborting thread (backtrace_symbols_fd: no allocation),
// to stderr and, when the variable's value is a path, appended to that file as well.  Both then hand over to whatever was installed before
// them (faulthandler's dump, the default action), so the process ends exactly as it would have.
#include <cxxabi.h>
#include <dirent.h>
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <typeinfo>

namespace {

int g_fd = -1;                       // the file copy (or -1), opened at the first line written: open(2) is async-signal-safe
char g_path[512];
struct sigaction g_prev;             // what handled SIGABRT before us
std::terminate_handler g_prev_term = nullptr;

void put(const char* s) {
    size_t n = strlen(s);
    if (g_fd < 0 && g_path[0]) g_fd = open(g_path, O_WRONLY | O_CREAT | O_APPEND, 0644);
    (void)!write(2, s, n);
    if (g_fd >= 0) (void)!write(g_fd, s, n);
}

void frames() {
    void* pc[96];
    int n = backtrace(pc, 96);
    backtrace_symbols_fd(pc, n, 2);
    if (g_fd >= 0) backtrace_symbols_fd(pc, n, g_fd);
}

// Every other thread of the process prints its own native frames (round 6: a process that is ended from outside because it STOPPED -- the
// test watchdog's SIGABRT to the blocked main thread, tests/conftest.py -- must say where each thread was: the runtime's own threads included).
volatile sig_atomic_t g_thread_done = 0;

void on_dump(int) {
    char head[96];
    snprintf(head, sizeof head, "[np abort] native frames of thread %ld:\n", (long)syscall(SYS_gettid));
    put(head);
    frames();
    g_thread_done = 1;
}

void dump_other_threads() {
    struct sigaction sa, old;
    memset(&sa, 0, sizeof sa);
    sa.sa_handler = on_dump;
    sigemptyset(&sa.sa_mask);
    if (sigaction(SIGUSR2, &sa, &old) != 0) return;
    const long me = (long)syscall(SYS_gettid);
    const int pid = (int)getpid();
    DIR* d = opendir("/proc/self/task");
    if (d) {
        int seen = 0;
        while (dirent* e = readdir(d)) {
            const long tid = atol(e->d_name);
            if (tid <= 0 || tid == me || ++seen > 256) continue;
            g_thread_done = 0;
            if (syscall(SYS_tgkill, pid, (int)tid, SIGUSR2) != 0) continue;
            for (int k = 0; k < 300 && !g_thread_done; ++k) usleep(1000);      // (a thread blocked inside the kernel never answers: 0.3 s, then the next)
            if (!g_thread_done) {
                char line[96];
                snprintf(line, sizeof line, "[np abort] thread %ld did not answer (blocked in the kernel?)\n", tid);
                put(line);
            }
        }
        closedir(d);
    }
    sigaction(SIGUSR2, &old, nullptr);
}

void on_abort(int sig) {
    alarm(60);      // (should the unwinder block on a lock the aborting thread holds, SIGALRM ends the process instead of a hang)
    put("[np abort] SIGABRT; native frames of the aborting thread:\n");
    frames();
    const char* all = getenv("NP_ABORT_TRACE_ALL");
    if (all && all[0] && all[0] != '0') dump_other_threads();
    sigaction(SIGABRT, &g_prev, nullptr);     // faulthandler's (or the default): the signal is raised again below and handled there
    raise(sig);
}

void on_terminate() {
    put("[np abort] std::terminate");
    if (std::type_info* t = abi::__cxa_current_exception_type()) {
        put(" with an exception in flight: ");
        put(t->name());
        try {
            throw;
        } catch (const std::exception& e) {
            put(": ");
            put(e.what());
        } catch (...) {
        }
    }
    put("\n");
    if (g_prev_term && g_prev_term != on_terminate) g_prev_term();
    abort();
}

struct Install {
    Install() {
        const char* v = getenv("NP_ABORT_TRACE");
        if (!v || !*v || !strcmp(v, "0")) return;
        // the other library of the pair was first in THIS process (each holds its own copy of this unit).  The marker carries the pid: a
        // child process inherits the environment and must still install its own handlers
        char me[32];
        snprintf(me, sizeof me, "%d", (int)getpid());
        const char* on = getenv("NP_ABORT_TRACE_ON");
        if (on && !strcmp(on, me)) return;
        setenv("NP_ABORT_TRACE_ON", me, 1);
        // a path ending in a pid that is not ours (inherited from the parent) gets ours appended, so every process writes its own file
        if (strchr(v, '/') && strlen(v) + 16 < sizeof g_path) {
            strcpy(g_path, v);
            // (exactly a trailing ".<pid>" or "_<pid>[.ext]" component of ours: a pid that is merely a substring of the path does not count)
            char dot[40], us[40];
            snprintf(dot, sizeof dot, ".%s", me);
            snprintf(us, sizeof us, "_%s.", me);
            const size_t lv = strlen(v), ld = strlen(dot);
            const bool mine = (lv >= ld && !strcmp(v + lv - ld, dot)) || strstr(v, us);
            if (!mine) { strcat(g_path, "."); strcat(g_path, me); }
        }
        void* warm[4];
        backtrace(warm, 4);                   // loads libgcc's unwinder now, not inside the handler
        struct sigaction sa;
        memset(&sa, 0, sizeof sa);
        sa.sa_handler = on_abort;
        sigemptyset(&sa.sa_mask);
        sa.sa_flags = SA_NODEFER;
        sigaction(SIGABRT, &sa, &g_prev);
        g_prev_term = std::set_terminate(on_terminate);
    }
} g_install;

}  // namespace
