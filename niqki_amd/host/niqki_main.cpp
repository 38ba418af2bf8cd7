// niqki_main.cpp -- the `niqki` command line on the MI355X engine.  Same options, output
// files, info box text and exit codes as the reference's program (behaviour of
// src/niqki.cpp:229-456, option table :102-185); here the run is a table of phases -- index
// inputs, dump, matrix, query inputs -- each a driver of nqhost::Index -> libniqki_hip.so.  Extensions (long options only):
//   --device <n>   HIP device ordinal (default: current device)
//   --gpus <n>     cut the index by sketch-slot range over n GPUs (devices --device .. +n-1)
//   --resident-mib <n>  paged index: sketch store in host memory, n MiB of device memory for one page of slots
#include <libgen.h>
#include <limits.h>
#include <unistd.h>
#include <fcntl.h>

#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "index_host.h"

using namespace std;
using namespace std::chrono;

namespace {

enum Opt { LIST, QUERY, LISTLINES, QUERYLINES, KMER, FETCH, OUTPUT, MIN, PRETTY, MATRIX, WORD, GENOME_SIZE, HHL,
           DUMP, LOAD, DOWNLAD, LOGO, HELP, DEVICE, GPUS, RESIDENT, N_OPT };
enum ArgKind { NONE, NONEMPTY, NUMERIC };

// Same order as the reference's descriptor table: a short option character
// selects the FIRST entry whose short-option string contains it
// (src/optionparser.h:1654), so -d -> indexdownload, -l -> querylines.
struct Desc { Opt id; const char *shorts; const char *longname; ArgKind kind; const char *help; };
const Desc kDesc[] = {
    {LIST, "I", "index", NONEMPTY, "  --index, -I <filename>        Input file of files to Index."},
    {QUERY, "Q", "query", NONEMPTY, "  --query, -Q <filename>        Input file of file to Query."},
    {LISTLINES, "i", "indexlines", NONEMPTY, "  --indexlines, -i <filename>   Query fa/fq file where each line is a separate entry to Index"},
    {QUERYLINES, "l", "querylines", NONEMPTY, "  --querylines, -q <filename>   Input fa/fq where each line is a separate entry to Query"},
    {KMER, "K", "kmer", NUMERIC, "  --kmer, -K <int>              Kmer size (31)."},
    {FETCH, "S", "sketch", NUMERIC, "  --sketch, -S <int>            Set sketch size to 2^S (15)."},
    {OUTPUT, "O", "output", NONEMPTY, "  --output, -O <filename>       Output file (niqkiOutput.gz)"},
    {MIN, "J", "minjac", NONEMPTY, "  --minjac, -J <int>            Minimal jaccard Index to report (0.1)."},
    {PRETTY, "P", "pretty", NONE, "  --pretty, -P                  Print a human-readable outfile. By default the outfile is in binary."},
    {MATRIX, "M", "matrix", NONEMPTY, "  --matrix, -M <filename>       Output the matrix distance to the given file."},
    {WORD, "W", "word", NUMERIC, "  --word, -W <int>              Fingerprint size (12)."},
    {GENOME_SIZE, "G", "Genomes_sizes", NUMERIC, "  --Genomes_sizes, -G <int>     Rought expectation of the genome sizes."},
    {HHL, "H", "HHL", NUMERIC, "  --HHL, -H <int>               Size of the hyperloglog section (4)."},
    {DUMP, "D", "dump", NONEMPTY, "  --dump, -D <filename>         Dump the current index to the given file."},
    {LOAD, "L", "load", NONEMPTY, "  --load, -L <filename>         Load an index to the given file."},
    {DOWNLAD, "Iddl", "indexdownload", NONEMPTY, "  --indexdownload, -Iddl <filename>  NCBI download (not available in this build)."},
    {LOGO, "", "logo", NONE, "  --logo                        Print ASCII art logo, then exit."},
    {HELP, "h", "help", NONE, "  --help, -h                    Print usage and exit."},
    {DEVICE, "", "device", NUMERIC, "  --device <int>                HIP device ordinal."},
    {GPUS, "", "gpus", NUMERIC, "  --gpus <int>                  Number of GPUs the index is sharded over (1)."},
    {RESIDENT, "", "resident-mib", NUMERIC, "  --resident-mib <int>          Device memory budget of a paged index in MiB (0: everything resident)."},
};

struct Parsed {
  bool error = false;
  map<int, vector<string>> opts;  // id -> args in order of appearance ("" for flag options)
  vector<string> non_options;
  bool has(Opt o) const { return opts.count(o) != 0; }
  const string &last(Opt o) const { return opts.at(o).back(); }
};

bool check_arg(const Desc &d, const char *arg, const string &shown) {
  if (d.kind == NONEMPTY) {
    if (arg && arg[0]) return true;
    fprintf(stderr, "Option '%s' requires a non-empty argument\n", shown.c_str());
    return false;
  }
  if (d.kind == NUMERIC) {
    char *end = nullptr;
    if (arg) (void)strtol(arg, &end, 10);
    if (arg && end != arg && *end == 0) return true;
    fprintf(stderr, "Option '%s' requires a numeric argument\n", shown.c_str());
    return false;
  }
  return true;
}

Parsed parse(int argc, char **argv) {
  Parsed p;
  int i = 0;
  for (; i < argc; ++i) {
    const char *a = argv[i];
    if (a[0] != '-' || a[1] == 0) break;            // first non-option ends option parsing (POSIX mode)
    if (a[1] == '-' && a[2] == 0) { ++i; break; }   // "--"
    if (a[1] == '-') {                              // long option
      string name = a + 2, val;
      bool attached = false;
      size_t eq = name.find('=');
      if (eq != string::npos) { val = name.substr(eq + 1); name = name.substr(0, eq); attached = true; }
      const Desc *d = nullptr;
      for (const auto &x : kDesc) if (name == x.longname) { d = &x; break; }
      if (!d) { fprintf(stderr, "Unknown option '%s'\n", name.c_str()); p.error = true; return p; }
      if (d->kind == NONE) { p.opts[d->id].push_back(""); continue; }
      const char *arg = attached ? val.c_str() : (i + 1 < argc ? argv[i + 1] : nullptr);
      if (!check_arg(*d, arg, name)) { p.error = true; return p; }
      p.opts[d->id].push_back(arg);
      if (!attached) ++i;
      continue;
    }
    for (const char *c = a + 1; *c; ++c) {          // short option group
      const Desc *d = nullptr;
      for (const auto &x : kDesc) if (x.shorts[0] && strchr(x.shorts, *c)) { d = &x; break; }
      string shown(1, *c);
      if (!d) { fprintf(stderr, "Unknown option '%s'\n", shown.c_str()); p.error = true; return p; }
      if (d->kind == NONE) { p.opts[d->id].push_back(""); continue; }
      const bool attached = c[1] != 0;
      const char *arg = attached ? c + 1 : (i + 1 < argc ? argv[i + 1] : nullptr);
      if (!check_arg(*d, arg, shown)) { p.error = true; return p; }
      p.opts[d->id].push_back(arg);
      if (!attached) ++i;
      break;  // the argument swallowed the rest of the group
    }
  }
  for (; i < argc; ++i) p.non_options.push_back(argv[i]);
  return p;
}

void print_usage() {
  clog << "\n***Input***\n";
  for (const auto &d : kDesc) {
    if (d.id == KMER) clog << "\n***Main parameters***\n";
    if (d.id == OUTPUT) clog << "\n***Output***\n";
    if (d.id == WORD) clog << "\n***Advanced parameters*** (You know what you are doing)\n";
    if (d.id == DUMP) clog << "\n***Index files***\n";
    if (d.id == DOWNLAD) clog << "\n***Other***\n";
    clog << d.help << "\n";
  }
}

// The names inside a file of files are relative to the list: while such a list is read the process
// sits in the list's directory (what src/niqki.cpp:200-224 does around its index and matrix phases).
class ListDir {
 public:
  explicit ListDir(const string &list_path) : back_(open(".", O_CLOEXEC)) {
    vector<char> path(list_path.begin(), list_path.end());
    path.push_back('\0');
    errno = 0;
    if (chdir(dirname(path.data())) != 0) cout << "Error: " << strerror(errno) << endl;
  }
  ~ListDir() {
    errno = 0;
    if (fchdir(back_) != 0) cout << "Error: " << strerror(errno) << endl;
    if (back_ >= 0) close(back_);
  }
  ListDir(const ListDir &) = delete;
  ListDir &operator=(const ListDir &) = delete;

 private:
  int back_;
};

string leaf_of(const string &path) { return path.substr(path.find_last_of("/\\") + 1); }

void warn_if_unreadable(const string &path) {
  if (!ifstream(path)) cout << "Unable to open the file '" << path << "'" << endl;
}

// the three time stamps behind the info box's "lasted" rows
struct RunClock {
  using tp = time_point<system_clock>;
  tp run_begin = system_clock::now(), index_end = run_begin;
  static void row(const char *label, tp from, tp to) {
    cout << label << setw(30) << setfill(' ') << duration<double>(to - from).count() << " |" << endl;
  }
  void index_done() {
    index_end = system_clock::now();
    row("| Indexing lasted (s)               |", run_begin, index_end);
  }
};

template <typename T>
void box_row(const char *label, const T &value) {
  cout << label << setw(30) << setfill(' ') << value << " |" << endl;
}

// One input option = one phase: which Index driver takes the file, and whether it is a list whose
// entries are relative to its own directory.
struct Phase {
  Opt opt;
  bool in_list_dir;
  void (nqhost::Index::*driver)(const string &);
};
const Phase kIndexPhases[] = {{LIST, true, &nqhost::Index::insert_file_of_file_whole},
                              {LISTLINES, true, &nqhost::Index::insert_file_lines}};
const Phase kQueryPhases[] = {{QUERY, false, &nqhost::Index::query_file_of_file_whole},
                              {QUERYLINES, false, &nqhost::Index::query_file_lines}};

void run_phase(nqhost::Index &ix, const Parsed &o, const Phase &ph) {
  if (!o.has(ph.opt)) return;
  const string path = o.last(ph.opt);
  warn_if_unreadable(path);
  if (ph.in_list_dir) {
    ListDir here(path);
    (ix.*ph.driver)(leaf_of(path));
  } else {
    (ix.*ph.driver)(path);
  }
}

// -M: the list doubles as the index input when no index option was given; the matrix itself is timed
// as a query of its own (both "lasted" rows are printed again, as the reference does)
void run_matrix(nqhost::Index &ix, const Parsed &o, RunClock &clk) {
  if (!o.has(MATRIX)) return;
  const string list = o.last(MATRIX);
  warn_if_unreadable(list);
  if (!o.has(LIST) && !o.has(LISTLINES)) {
    clk.run_begin = system_clock::now();
    {
      ListDir here(list);
      ix.insert_file_of_file_whole(leaf_of(list));
    }
    clk.index_done();
  }
  ListDir here(list);
  clk.run_begin = system_clock::now();
  ix.query_matrix();
  RunClock::row("| Query lasted (s)                  |", clk.run_begin, system_clock::now());
}

int int_opt(const Parsed &o, Opt id, int dflt) { return o.has(id) ? atoi(o.last(id).c_str()) : dflt; }

}  // namespace

int main(int argc, char *argv[]) {
  if (argc > 0) { --argc; ++argv; }
  const Parsed o = parse(argc, argv);
  if (o.error) {
    cout << "Bad usage!!!" << endl;
    return EXIT_FAILURE;
  }
  if (o.has(HELP) || argc == 0) {
    print_usage();
    return EXIT_SUCCESS;
  }
  for (size_t i = 0; i < o.non_options.size(); ++i)
    cout << "Non-option argument #" << i << " is " << o.non_options[i] << endl
         << "Ignoring unknown argument '" << o.non_options[i] << "'" << endl;
  if (!o.non_options.empty()) {
    cout << "Bad usage!!!" << endl;
    return EXIT_FAILURE;
  }
  const int K = int_opt(o, KMER, 31), S = int_opt(o, FETCH, 15), H = int_opt(o, HHL, 4), W = int_opt(o, WORD, 12);
  const double min_jaccard = o.has(MIN) ? atof(o.last(MIN).c_str()) : 0;
  const int device = int_opt(o, DEVICE, -1), n_gpus = int_opt(o, GPUS, 1), resident_mib = int_opt(o, RESIDENT, 0);
  const string out_file = o.has(OUTPUT) ? o.last(OUTPUT) : "niqkiOutput.gz";

  const char *rule = "+-----------------------------------+-------------------------------+";
  cout << "+-------------------------------------------------------------------+" << endl
       << "|                            Informations                           |" << endl
       << rule << endl;
  std::unique_ptr<nqhost::Index> ix;
  try {
    if (o.has(LOAD)) ix.reset(new nqhost::Index(o.last(LOAD), true, out_file, device, n_gpus, resident_mib));
    else ix.reset(new nqhost::Index(S, K, W, H, out_file, min_jaccard, device, n_gpus, resident_mib));
    if (const unsigned expect = (unsigned)int_opt(o, GENOME_SIZE, 0)) ix->select_best_H(expect);   // src/niqki.cpp:303-305

    RunClock clk;
    for (const Phase &ph : kIndexPhases) run_phase(*ix, o, ph);
    if (o.has(DOWNLAD)) cout << "--indexdownload needs network access and is not part of this build" << endl;
    if (o.has(DUMP)) ix->dump_index_disk(o.last(DUMP));
    clk.index_done();
    run_matrix(*ix, o, clk);
    for (const Phase &ph : kQueryPhases) run_phase(*ix, o, ph);
    ix->outfile->close();
    const RunClock::tp done = system_clock::now();
    RunClock::row("| Query lasted (s)                  |", clk.index_end, done);
    RunClock::row("| Whole run lasted (s)              |", clk.run_begin, done);
  } catch (const std::exception &e) {
    cerr << "niqki: " << e.what() << endl;
    return EXIT_FAILURE;
  }

  if (o.has(LOGO)) {
    ifstream art("../resources/niqki.ascii");
    if (!art.is_open()) cout << "Unable to open file :'../resources/niqki.ascii'" << endl;
    for (string line; art.is_open() && getline(art, line);) cout << line << '\n';
    return EXIT_SUCCESS;
  }
  cout << rule << endl;
  box_row("| k-mer size                        |", K);
  box_row("| S                                 |", S);
  box_row("| Number of fingerprints            |", ix->F);
  box_row("| W                                 |", W);
  box_row("| H                                 |", H);
  box_row("| Number of indexed genomes         |", ix->getNbGenomes());
  cout << rule << endl;
  return EXIT_SUCCESS;
}
