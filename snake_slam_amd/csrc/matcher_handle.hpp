// The matcher / preprocess handle (shared by matcher.hip and preprocess.hip).
#pragma once
#include "common.hpp"

struct snk_matcher : snk::HandleBase
{
    snk::DevBuf q, t, out, aux, aux2, cnt;
    // device copy of the frame the projection matchers were last called with (track.hip): 1-2 coarse calls and one
    // fine call per frame (TrackingCoarse.cpp:234, TrackingFine.cpp:149) look at the same frame, which is uploaded once
    snk::DevBuf view;
    unsigned long long view_key[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool view_valid                = false;
};
