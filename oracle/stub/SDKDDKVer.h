// Empty stand-in for the Windows SDK header pulled in by the reference targetver.h:8.
#pragma once
