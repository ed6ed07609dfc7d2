// Case-fix forwarder: the reference includes "StdAfx.h" but ships "stdafx.h" (Windows FS is
// case-insensitive). Test infrastructure only (oracle build); see oracle/README.md.
#pragma once
#include "stdafx.h"
#include "er_oracle_stub.h"
