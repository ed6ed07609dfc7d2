// Case forwarder: the reference includes "StdAfx.h", the file is stdafx.h (Windows file systems do not care).
#pragma once
#include <cmath>
#include <cstring>
#include <string>
#include "../stub/er_oracle_stub.h"
#include "stdafx.h"
