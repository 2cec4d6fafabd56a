// lk_internal.h -- symbols shared between the translation units of liblightkrylov_hip.so (not part of the ABI).
#pragma once
#include "../../include/lightkrylov_hip.h"

// records the message returned by lk_last_error() and returns `code`
extern "C" int lk_fail_(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3), visibility("hidden")));
