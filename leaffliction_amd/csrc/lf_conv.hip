#include "lf_common.h"
