// niqki_main.cpp -- the `niqki` command line on the MI355X engine.  Same options,
// phases, output files and info box as the reference's main()
// (src/niqki.cpp:229-456, option table :102-185); the work is done by
// nqhost::Index -> libniqki_hip.so.  Extensions (long options only):
//   --device <n>   HIP device ordinal (default: current device)
//   --gpus <n>     cut the index by sketch-slot range over n GPUs (devices --device .. +n-1)
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
#include <string>
#include <vector>

#include "index_host.h"

using namespace std;
using namespace std::chrono;

namespace {

enum Opt { LIST, QUERY, LISTLINES, QUERYLINES, KMER, FETCH, OUTPUT, MIN, PRETTY, MATRIX, WORD, GENOME_SIZE, HHL,
           DUMP, LOAD, DOWNLAD, LOGO, HELP, DEVICE, GPUS, N_OPT };
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

int old_wd = -1;
void changeDirFromFilename(const char *fname) {  // src/niqki.cpp:202-215
  old_wd = open(".", O_CLOEXEC);
  char copy[PATH_MAX];
  strncpy(copy, fname, PATH_MAX);
  copy[PATH_MAX - 1] = '\0';
  errno = 0;
  if (chdir(dirname(copy))) cout << "Error: " << strerror(errno) << endl;
}
void restoreDir() {  // src/niqki.cpp:219-224
  errno = 0;
  if (fchdir(old_wd)) cout << "Error: " << strerror(errno) << endl;
}
string base_name(const string &s) { return s.substr(s.find_last_of("/\\") + 1); }

void box_time(const char *label, double s) {
  cout << label << setw(30) << setfill(' ') << s << " |" << endl;
}

}  // namespace

int main(int argc, char *argv[]) {
  argc -= (argc > 0);
  argv += (argc > 0);
  Parsed o = parse(argc, argv);
  if (o.error) {
    cout << "Bad usage!!!" << endl;
    return EXIT_FAILURE;
  }
  if (o.has(HELP) || argc == 0) {
    print_usage();
    return EXIT_SUCCESS;
  }
  const int K = o.has(KMER) ? atoi(o.last(KMER).c_str()) : 31;
  const int F = o.has(FETCH) ? atoi(o.last(FETCH).c_str()) : 15;
  const int H = o.has(HHL) ? atoi(o.last(HHL).c_str()) : 4;
  const int W = o.has(WORD) ? atoi(o.last(WORD).c_str()) : 12;
  const double min_fract = o.has(MIN) ? atof(o.last(MIN).c_str()) : 0;
  const unsigned genomes_sizes = o.has(GENOME_SIZE) ? (unsigned)atoi(o.last(GENOME_SIZE).c_str()) : 0;
  const int device = o.has(DEVICE) ? atoi(o.last(DEVICE).c_str()) : -1;
  const int n_gpus = o.has(GPUS) ? atoi(o.last(GPUS).c_str()) : 1;

  for (size_t i = 0; i < o.non_options.size(); ++i) {
    cout << "Non-option argument #" << i << " is " << o.non_options[i] << endl;
    cout << "Ignoring unknown argument '" << o.non_options[i] << "'" << endl;
  }
  if (!o.non_options.empty()) {
    cout << "Bad usage!!!" << endl;
    return EXIT_FAILURE;
  }
  const string out_file = o.has(OUTPUT) ? o.last(OUTPUT) : "niqkiOutput.gz";
  cout << "+-------------------------------------------------------------------+" << endl;
  cout << "|                            Informations                           |" << endl;
  cout << "+-----------------------------------+-------------------------------+" << endl;
  nqhost::Index *monindex = nullptr;
  try {
    if (o.has(LOAD)) monindex = new nqhost::Index(o.last(LOAD), true, out_file, device, n_gpus);
    else monindex = new nqhost::Index(F, K, W, H, out_file, min_fract, device, n_gpus);
  } catch (const std::exception &e) {
    cerr << "niqki: " << e.what() << endl;
    return EXIT_FAILURE;
  }
  if (genomes_sizes != 0) {
    try {
      monindex->select_best_H(genomes_sizes);  // src/niqki.cpp:303-305
    } catch (const std::exception &e) {
      cerr << "niqki: " << e.what() << endl;
      return EXIT_FAILURE;
    }
  }

  time_point<system_clock> start, endindex, end;
  start = system_clock::now();
  try {
    if (o.has(LIST)) {
      const string list_file = o.last(LIST);
      ifstream ifs(list_file);
      if (!ifs) cout << "Unable to open the file '" << list_file << "'" << endl;
      changeDirFromFilename(list_file.c_str());
      monindex->insert_file_of_file_whole(base_name(list_file));
      restoreDir();
    }
    if (o.has(LISTLINES)) {
      const string list_file = o.last(LISTLINES);
      ifstream ifs(list_file);
      if (!ifs) cout << "Unable to open the file '" << list_file << "'" << endl;
      changeDirFromFilename(list_file.c_str());
      monindex->insert_file_lines(base_name(list_file));
      restoreDir();
    }
    if (o.has(DOWNLAD))
      cout << "--indexdownload needs network access and is not part of this build" << endl;
    if (o.has(DUMP)) monindex->dump_index_disk(o.last(DUMP));

    endindex = system_clock::now();
    duration<double> elapsed = endindex - start;
    box_time("| Indexing lasted (s)               |", elapsed.count());

    if (o.has(MATRIX)) {
      const string matrix_file = o.last(MATRIX);
      ifstream ifs(matrix_file);
      if (!ifs) cout << "Unable to open the file '" << matrix_file << "'" << endl;
      if (!o.has(LIST) && !o.has(LISTLINES)) {
        start = system_clock::now();
        changeDirFromFilename(matrix_file.c_str());
        monindex->insert_file_of_file_whole(base_name(matrix_file));
        restoreDir();
        endindex = system_clock::now();
        elapsed = endindex - start;
        box_time("| Indexing lasted (s)               |", elapsed.count());
      }
      changeDirFromFilename(matrix_file.c_str());
      start = system_clock::now();
      monindex->query_matrix();
      end = system_clock::now();
      elapsed = end - start;
      box_time("| Query lasted (s)                  |", elapsed.count());
      restoreDir();
    }
    if (o.has(QUERY)) {
      const string query_file = o.last(QUERY);
      ifstream ifs(query_file);
      if (!ifs) cout << "Unable to open the file '" << query_file << "'" << endl;
      monindex->query_file_of_file_whole(query_file);
    }
    if (o.has(QUERYLINES)) {
      const string query_file = o.last(QUERYLINES);
      ifstream ifs(query_file);
      if (!ifs) cout << "Unable to open the file '" << query_file << "'" << endl;
      monindex->query_file_lines(query_file);
    }
    monindex->outfile->close();
  } catch (const std::exception &e) {
    cerr << "niqki: " << e.what() << endl;
    return EXIT_FAILURE;
  }

  end = system_clock::now();
  duration<double> elapsed = end - endindex;
  box_time("| Query lasted (s)                  |", elapsed.count());
  elapsed = end - start;
  box_time("| Whole run lasted (s)              |", elapsed.count());

  if (o.has(LOGO)) {
    ifstream logo("../resources/niqki.ascii");
    string line;
    if (logo.is_open()) while (getline(logo, line)) cout << line << '\n';
    else cout << "Unable to open file :'../resources/niqki.ascii'" << endl;
    return EXIT_SUCCESS;
  }
  cout << "+-----------------------------------+-------------------------------+" << endl;
  cout << "| k-mer size                        |" << setw(30) << setfill(' ') << K << " |" << endl
       << "| S                                 |" << setw(30) << setfill(' ') << F << " |" << endl
       << "| Number of fingerprints            |" << setw(30) << setfill(' ') << monindex->F << " |" << endl
       << "| W                                 |" << setw(30) << setfill(' ') << W << " |" << endl
       << "| H                                 |" << setw(30) << setfill(' ') << H << " |" << endl
       << "| Number of indexed genomes         |" << setw(30) << setfill(' ') << monindex->getNbGenomes() << " |" << endl;
  cout << "+-----------------------------------+-------------------------------+" << endl;
  delete monindex;
  return EXIT_SUCCESS;
}
