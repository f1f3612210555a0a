// thread-local last-error string shared by every translation unit of librectorch_hip
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[1024] = "";

void rtx_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

const char* rtx_last_error_str() { return g_err; }
