/* Forward declaration ONLY (no behaviour): PT_sv5_/Model.cpp:28-43 defines std::operator<(tinyobj::index_t, ...)
 * after <map> has been included and relies on MSVC finding it at instantiation time; a conforming compiler
 * needs the declaration to be visible before std::less is defined.  Force-included by `make ref` for Model.cpp. */
#pragma once
#include <math.h>
namespace tinyobj { struct index_t; }
namespace std { inline bool operator<(const tinyobj::index_t& a, const tinyobj::index_t& b); }
