// rs_handle.hpp -- host-side handle shared by the translation units of librs_hip.so
#pragma once
#include "../../include/radsearch.h"
#include "rs_device.hpp"

struct rs_field { const char* name; void* ptr; int elem, rows, cols; };

struct rs_handle {
    rs_config cfg;
    RsParams P;
    int device;
    size_t bytes;
    int n_fields;
    rs_field fields[32];
};
