// makestat_main.cpp -- native stand-in for the reference's bin/make.stat.pl: the consumer of the
// <sid>.<mode>2pairs.log files this package writes.  Same argv (`makestat <sid> <concat=yes|no>`), same inputs
// (<sid>.trim.log, <sid>.rmdup.log, <sid>.stitch.stat | <sid>.flash.log [+ <sid>.cut.log], <sid>.flash2pairs.log,
// <sid>.unc2pairs.log), byte-identical <sid>.final.stat text on stdout (make.stat.pl:21-130), usage exit code 2.
// Host-only text processing of a dozen numbers: nothing here runs on the GPU.
// Extension (never by default): `--chrstat` appends the per-chromosome-pair contact counts of <sid>.unc.chrstat /
// <sid>.flash.chrstat (written by `sam2pairs` under MKT_EXT=1) as an extra section, leaving the legacy lines alone.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <regex>
#include <sstream>
#include <string>
#include <vector>

typedef std::map<std::string, double> Table;

static bool load_kv(const std::string& path, Table& t, bool also_all = false) {      // "key \t value" lines, summed per key
    std::ifstream in(path.c_str());
    if (in.fail()) return false;
    std::string line;
    while (std::getline(in, line)) {
        const size_t tab = line.find('\t');
        if (tab == std::string::npos) continue;
        const size_t tab2 = line.find('\t', tab + 1);
        const double v = atof(line.substr(tab + 1, tab2 == std::string::npos ? std::string::npos : tab2 - tab - 1).c_str());
        t[line.substr(0, tab)] += v;
        if (also_all) t["all"] += v;
    }
    return true;
}
// make.stat.pl:132-136: thousands separators
static std::string d(double v) {
    char buf[64];
    snprintf(buf, sizeof buf, "%.0f", v);
    std::string s = buf, out;
    const bool neg = !s.empty() && s[0] == '-';
    const std::string digits = neg ? s.substr(1) : s;
    for (size_t i = 0; i < digits.size(); ++i) {
        out += digits[i];
        const size_t left = digits.size() - 1 - i;
        if (left && left % 3 == 0) out += ',';
    }
    return neg ? "-" + out : out;
}
static void die(const std::string& what) { fprintf(stderr, "%s\n", what.c_str()); exit(2); }
static double nz(double x) { if (x == 0) die("Illegal division by zero"); return x; }

int main(int argc, char* argv[]) {
    std::vector<std::string> pos;
    bool chrstat = false;
    for (int i = 1; i < argc; ++i) { if (!strcmp(argv[i], "--chrstat")) chrstat = true; else pos.push_back(argv[i]); }
    if (pos.size() < 2) { fprintf(stderr, "\nUsage: %s <sid> <concat=yes|no>\n\n", argv[0]); return 2; }
    const std::string sid = pos[0], concat = pos[1];
    printf("#Category\tCount\tFraction(%%)\n");
    Table trim, rmdup;
    if (!load_kv(sid + ".trim.log", trim)) die("cat: " + sid + ".trim.log: No such file or directory");
    if (!load_kv(sid + ".rmdup.log", rmdup)) die("No such file or directory: " + sid + ".rmdup.log");
    printf("## Preprocessing and alignment\n");
    printf("Total\t%s\t100.0\nKtrim\t%s\t%.1f\nUnique\t%s\t%.1f\n", d(trim["Total"]).c_str(), d(rmdup["Total"]).c_str(),
           rmdup["Total"] / nz(trim["Total"]) * 100, d(rmdup["Uniq"]).c_str(), rmdup["Uniq"] / nz(rmdup["Total"]) * 100);
    double prealign;
    if (concat == "yes") {
        double cat = 0, unc = 0, cut = 0;
        std::ifstream fl((sid + ".flash.log").c_str());
        fl.seekg(0, std::ios::end);
        if (!fl.fail() && fl.tellg() > 0) {          // old version (make.stat.pl:52-74)
            fl.seekg(0);
            std::string line;
            const std::regex rc("\\sCombined pairs:\\s+(\\d+)");
            std::smatch m;
            while (std::getline(fl, line)) if (std::regex_search(line, m, rc)) { cat = atof(m[1].str().c_str()); break; }
            std::ifstream fc((sid + ".cut.log").c_str());
            fc.seekg(0, std::ios::end);
            if (!fc.fail() && fc.tellg() > 0) {
                fc.seekg(0);
                const std::regex rt("Total\\s+(\\d+)"), rp("Pass\\s+(\\d+)");
                while (std::getline(fc, line)) {
                    if (std::regex_search(line, m, rt)) unc = atof(m[1].str().c_str());
                    if (std::regex_search(line, m, rp)) cut = atof(m[1].str().c_str());
                }
            } else { unc = rmdup["Uniq"] - cat; cut = unc; }
        } else {                                      // new version (make.stat.pl:75-82)
            std::ifstream fs((sid + ".stitch.stat").c_str());
            if (fs.fail()) die("No such file or directory: " + sid + ".stitch.stat");
            std::string line;
            std::getline(fs, line);
            std::vector<std::string> l;
            std::stringstream ss(line);
            std::string f;
            while (std::getline(ss, f, '\t')) l.push_back(f);
            l.resize(6);
            cat = atof(l[1].c_str()); unc = atof(l[3].c_str()); cut = atof(l[5].c_str());
        }
        printf("Stitched\t%s\t%.1f\nUnstitched\t%s\t%.1f\n  Discarded(too-short)\t%s\t%.1f\n", d(cat).c_str(), cat / nz(rmdup["Uniq"]) * 100,
               d(cut).c_str(), cut / nz(rmdup["Uniq"]) * 100, d(unc - cut).c_str(), (unc - cut) / nz(rmdup["Uniq"]) * 100);
        prealign = cat + cut;
    } else prealign = rmdup["Uniq"];
    Table align;
    if (concat == "yes" && !load_kv(sid + ".flash2pairs.log", align, true)) die("No such file or directory: " + sid + ".flash2pairs.log");
    if (!load_kv(sid + ".unc2pairs.log", align, true)) die("No such file or directory: " + sid + ".unc2pairs.log");
    const double all = align["all"];
    printf("Mappable\t%s\t%.1f\n", d(all).c_str(), all / nz(prealign) * 100);
    printf("## Interactions\n");
    const double uncalled = align["lowMap"] + align["manyHits"] + align["unpaired"] + align["selfCircle"];
    printf("Uncalled\t%s\t%.1f\n", d(uncalled).c_str(), uncalled / nz(all) * 100);
    printf("  Incomplete-mapping\t%s\t%.1f\n", d(align["lowMap"]).c_str(), align["lowMap"] / all * 100);
    printf("  Too-many-segments\t%s\t%.1f\n", d(align["manyHits"]).c_str(), align["manyHits"] / all * 100);
    printf("  Unpairable\t%s\t%.1f\n", d(align["unpaired"]).c_str(), align["unpaired"] / all * 100);
    printf("  Self-circle\t%s\t%.1f\n", d(align["selfCircle"]).c_str(), align["selfCircle"] / all * 100);
    const double valid = align["trans"] + align["cis10K"] + align["cis1K"] + align["cis0"];
    printf("Reported\t%s\t%.1f\n", d(valid).c_str(), valid / all * 100);
    printf("  Cis(<1K)\t%s\t%.1f\n", d(align["cis0"]).c_str(), align["cis0"] / all * 100);
    printf("  Cis(1-10K)\t%s\t%.1f\n", d(align["cis1K"]).c_str(), align["cis1K"] / all * 100);
    printf("  Cis(>=10K)\t%s\t%.1f\n", d(align["cis10K"]).c_str(), align["cis10K"] / all * 100);
    printf("  Trans\t%s\t%.1f\n", d(align["trans"]).c_str(), align["trans"] / all * 100);
    if (chrstat) {                                    // extension: never part of the legacy table
        std::map<std::pair<std::string, std::string>, double> cc;
        for (const char* mode : {"flash", "unc"}) {
            std::ifstream in((sid + "." + mode + ".chrstat").c_str());
            std::string a, b;
            double v;
            while (in >> a >> b >> v) cc[std::make_pair(a, b)] += v;
        }
        printf("## Contacts per chromosome pair\n");
        for (const auto& kv : cc) printf("%s\t%s\t%s\n", kv.first.first.c_str(), kv.first.second.c_str(), d(kv.second).c_str());
    }
    return 0;
}
