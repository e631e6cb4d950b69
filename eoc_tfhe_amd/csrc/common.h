// common.h -- small shared helpers of the engine and the host-side client code.
#pragma once
#include <cstdarg>
#include <cstddef>
#include <cstdio>

// last error message (also echoed on stderr, the reference's convention:
// ao-tfhe/eoc-tfhe-run.cpp:218-219,277-278)
void eoc_set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
// the message is per thread (eoc_last_error reads the calling thread's): a result produced on a worker thread hands its
// message over to the thread that reports the error code (already echoed on stderr once, so not again)
void eoc_adopt_error(const char *msg);

// levels of a netlist as eoc_circuit_run_device evaluates it (host.cpp): RAW, WAR and WAW hazards on wires, 1-based;
// gates must have been checked (valid opcodes, wire ids below n_wires).  Returns the number of levels.
struct eoc_gate;
int eoc_levelise(const eoc_gate *gates, size_t n_gates, size_t n_wires, int *level_of);
