#pragma once
#include "../../../stub/er_oracle_stub.h"
