// The replay ring's handle and its device-side counters, shared by csrc/replay.hip and csrc/step_path.hip.
#pragma once
#include "common.h"
#include "../../include/naf_hip.h"

struct naf_replay {
    uint64_t capacity;
    int S, A, row_floats;
    float* rows;
    uint64_t* meta;  // {head, size, total_added, sample_counter, -, -, -, bad_index_count}
    uint32_t magic;
};
#define NAF_REPLAY_MAGIC 0x4e414652u

enum { META_HEAD = 0, META_SIZE = 1, META_TOTAL = 2, META_SAMPLE_CTR = 3, META_BAD_IDX = 7 };

