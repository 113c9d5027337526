/* Minimal TCLAP look-alike (the argument classes srba-slam declares): "--name value" / "-f value" / switches. */
#pragma once
#include <mrpt_lite_apps.h> // (the reference reaches mrpt::system & co. through <srba.h>, which includes the MRPT umbrella headers)
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>
namespace TCLAP {
struct ArgException : public std::runtime_error { ArgException(const std::string &m) : std::runtime_error(m) {} std::string error() const { return what(); } std::string argId() const { return std::string(); } };
struct Arg { std::string flag, name, desc; bool set = false, is_switch = false; virtual ~Arg() {} virtual void take(const std::string &) = 0; bool isSet() const { return set; } };
class CmdLine {
public:
	CmdLine(const std::string &message, char = ' ', const std::string &version = "none", bool = true) : m_message(message), m_version(version) {}
	void add(Arg *a) { m_args.push_back(a); } void add(Arg &a) { m_args.push_back(&a); }
	bool parse(int argc, char **argv) {
		for (int i = 1; i < argc; i++) {
			const std::string s(argv[i]); Arg *hit = NULL;
			for (size_t k = 0; k < m_args.size(); k++) if (s == "--" + m_args[k]->name || (!m_args[k]->flag.empty() && s == "-" + m_args[k]->flag)) hit = m_args[k];
			if (!hit) throw ArgException("unknown argument: " + s);
			if (hit->is_switch) hit->take("1"); else { if (i + 1 >= argc) throw ArgException("missing value for " + s); hit->take(argv[++i]); }
		}
		return true;
	}
	const std::string &getMessage() const { return m_message; }
private:
	std::string m_message, m_version; std::vector<Arg *> m_args;
};
template <class T> class ValueArg : public Arg {
public:
	ValueArg(const std::string &flag_, const std::string &name_, const std::string &desc_, bool /*required*/, T def, const std::string & /*type*/, CmdLine &cmd) : m_value(def) { flag = flag_; name = name_; desc = desc_; cmd.add(this); }
	void take(const std::string &s) override { std::istringstream is(s); is >> m_value; set = true; }
	const T &getValue() const { return m_value; }
private:
	T m_value;
};
template <> inline void ValueArg<std::string>::take(const std::string &s) { m_value = s; set = true; }
class SwitchArg : public Arg {
public:
	SwitchArg(const std::string &flag_, const std::string &name_, const std::string &desc_, CmdLine &cmd, bool def = false) : m_value(def) { flag = flag_; name = name_; desc = desc_; is_switch = true; cmd.add(this); }
	void take(const std::string &) override { m_value = !m_value; set = true; }
	bool getValue() const { return m_value; }
private:
	bool m_value;
};
} // namespace TCLAP
