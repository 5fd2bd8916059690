// The matcher / preprocess handle (shared by matcher.hip and preprocess.hip).
#pragma once
#include "common.hpp"

struct snk_matcher : snk::HandleBase
{
    snk::DevBuf q, t, out, aux, aux2, cnt;
};
