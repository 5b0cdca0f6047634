#include "ParameterList.h"
