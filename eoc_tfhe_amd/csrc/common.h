// common.h -- small shared helpers of the engine and the host-side client code.
#pragma once
#include <cstdarg>
#include <cstdio>

// last error message (also echoed on stderr, the reference's convention:
// ao-tfhe/eoc-tfhe-run.cpp:218-219,277-278)
void eoc_set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
