// dr_options.h -- the library's tuning / diagnostic switches (DARTRAY_*), by VALUE.
//
// A switch is the value given to dr_set_option(name, value) of the C ABI, else the environment's; unset when neither
// exists (or dr_set_option hid the environment's with "").  It is read at every use -- nothing is latched at first use --
// so a long-lived foreign host can change a switch between two renders without setenv.
//
// Round 5: dr_opt returns a COPY.  (Rounds 3-4 handed out a pointer into one thread-local buffer: a second look-up
// overwrote what the first had returned, so `a = dr_option("STATE_LAYOUT"); dr_option("LAYOUT_PILOT"); atoi(a)` read
// the pilot switch's value whenever both had been set through dr_set_option.)  No caller can hold anything that a later
// look-up or a dr_set_option on another thread invalidates.
#ifndef DR_OPTIONS_H
#define DR_OPTIONS_H

#include <cstdlib>
#include <string>

struct DrOpt {
  bool set = false;
  std::string value;
  explicit operator bool() const { return set; }
  // the value as an integer (atoi semantics), `dflt` when the switch is unset
  int toInt(int dflt) const { return set ? atoi(value.c_str()) : dflt; }
  // "set and zero": the spelling of the off switches (DARTRAY_PILOT=0, DARTRAY_OVERLAP_ANY=0, ...)
  bool isZero() const { return set && atoi(value.c_str()) == 0; }
  bool nonZero() const { return set && atoi(value.c_str()) != 0; }
  bool is(const char* s) const { return set && value == s; }
  char first() const { return set && !value.empty() ? value[0] : '\0'; }
};

DrOpt dr_opt(const char* name);  // dr_api.hip

#endif
