// The kernel-level handle of include/dacapo_ckks.h: a (possibly borrowed) Context.
#pragma once
#include "context.hpp"

struct dc_context {
    dacapo::Context *c;
    bool owned; // false when the context belongs to an HEVM (hevm_context())
    void *item_ring = nullptr; // device slots for the one-item tables of the fused composite ops (c_api.hip)
    unsigned item_next = 0;
};
