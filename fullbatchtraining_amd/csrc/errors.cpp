#include "common.h"
thread_local char fb_err_buf[512] = "";
extern "C" const char* fb_last_error_string(void) { return fb_err_buf; }
extern "C" int fb_abi_version(void) { return 1; }
