// Internal glue: public ABI types + shared helpers.
#pragma once
#include "../../include/sscg.h"
