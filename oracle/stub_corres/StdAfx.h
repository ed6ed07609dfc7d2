// Case forwarder: the reference's CorresApp.cpp includes "StdAfx.h", the file is stdafx.h (Windows file systems do not care).
// TEST INFRASTRUCTURE (oracle build of /root/reference/BuildCorrespondence, see oracle/README.md).
#pragma once
#include "er_corres_stub.h"
#include "stdafx.h"
