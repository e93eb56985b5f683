// TEST INFRASTRUCTURE: forwards to the API stand-in (see velox_stub.h).
#pragma once
#include "velox_stub.h"
