// MOCK (see lmp_mock_core.h)
#pragma once
#include "lmp_mock_core.h"
