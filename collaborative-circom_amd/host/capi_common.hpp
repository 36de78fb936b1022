// shared by the C entry-point translation units: the thread-local error text and the per-party error report
#pragma once
#include <string>
extern thread_local std::string g_host_err;   // defined in capi_tools.cpp
// first real failure among the parties (the others only report that somebody else died)
template <class Errs> static bool report_party_errors(const Errs& errs, int n) {
    int pick = -1;
    for (int i = 0; i < n; i++) if (!errs[i].empty() && (pick < 0 || (errs[pick] == "another party failed" && errs[i] != "another party failed"))) pick = i;
    if (pick < 0) return false;
    g_host_err = "party " + std::to_string(pick) + ": " + errs[pick];
    return true;
}
