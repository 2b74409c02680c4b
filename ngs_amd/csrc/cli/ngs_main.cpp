// ngs_main.cpp -- the `ngs qc` command line over the MI355X hot path.
//
// Mirrors the reference's CLI surface for this subcommand (flag names, defaults, error texts,
// output file): src/main.rs:19-105 (global -q/-v, dispatch), src/qc/command.rs:36-102 (QcArgs),
// :109-218 (qc), :226-421 (app: open_and_parse, sequence concordance check, facet selection,
// pass 1 / pass 2 with the two `-n` rules, aggregate, write <prefix>.results.json).
// The per-record facet loops are replaced by SoA batches through the C ABI (include/ngsq.h);
// ingest is include/ngsq_bam.h.  Additive flags: --device, --batch-records, --threads, --gc-seed,
// --ingest host|device (default device: the GPU inflates and parses the BAM; runs with -n
// always ingest on the host), --coverage auto|stream|array, and --gpus N: one worker process per
// GPU, each ingesting its BGZF block range of the file on its own device, one ngsq_exchange
// (include/ngsq_comm.h; RCCL over xGMI) before the teardown, rank 0 writes the JSON.
// Not built (out of scope this round): the other subcommands.
#include <poll.h>
#include <signal.h>
#include <spawn.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <charconv>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <filesystem>
#include <fstream>
#include <map>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/ngsq.h"
#include "../../../include/ngsq_bam.h"
#include "../../../include/ngsq_stage.h"
#include "../../../include/ngsq_comm.h"
#include "../../../include/ngsq_reference.h"
#include "gff_loader.h"

namespace {

// cores the command may use: the cgroup's CPU quota when there is one (as the library's readers count them)
int cgroup_cores() {
    int n = (int)std::thread::hardware_concurrency();
    if (n < 1) n = 1;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char quota[32] = {0};
        long period = 0;
        if (fscanf(f, "%31s %ld", quota, &period) == 2 && strcmp(quota, "max") != 0 && period > 0) {
            const long q = atol(quota) / period;
            if (q >= 1 && q < n) n = (int)q;
        }
        fclose(f);
    }
    return n;
}

int g_level = 2; // 0 off (-q), 2 info (default), 3 debug (-v)   src/main.rs:71-83

void logf(int level, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
void logf(int level, const char *fmt, ...) {
    if (level > g_level) return;
    char ts[64];
    const auto now = std::chrono::system_clock::now();
    const std::time_t t = std::chrono::system_clock::to_time_t(now);
    std::tm tm{};
    gmtime_r(&t, &tm);
    const long us = (long)(std::chrono::duration_cast<std::chrono::microseconds>(now.time_since_epoch()).count() % 1000000);
    strftime(ts, sizeof ts, "%Y-%m-%dT%H:%M:%S", &tm);
    fprintf(stderr, "%s.%06ldZ %5s ngs::qc::command: ", ts, us, level <= 1 ? "ERROR" : level == 2 ? "INFO" : "DEBUG");
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fputc('\n', stderr);
}

[[noreturn]] void bail(const std::string &msg) { // anyhow::bail! -> "Error: ..." and exit code 1
    fprintf(stderr, "Error: %s\n", msg.c_str());
    if (ngsq_comm_rccl_stuck()) { // a thread is still inside ncclCommInitRank: the exit handlers may wait for it
        fflush(nullptr);
        _exit(1);
    }
    exit(1);
}

std::string with_commas(unsigned long long v) { // num_format Locale::en
    std::string s = std::to_string(v), out;
    for (size_t i = 0; i < s.size(); i++) {
        out += s[i];
        const size_t left = s.size() - 1 - i;
        if (left && left % 3 == 0) out += ',';
    }
    return out;
}

bool ieq(const std::string &a, const std::string &b) {
    if (a.size() != b.size()) return false;
    for (size_t i = 0; i < a.size(); i++)
        if (tolower((unsigned char)a[i]) != tolower((unsigned char)b[i])) return false;
    return true;
}

std::string exe_dir() {
    char buf[4096];
    const ssize_t n = readlink("/proc/self/exe", buf, sizeof buf - 1);
    if (n <= 0) return ".";
    buf[n] = 0;
    std::string p(buf);
    const size_t k = p.rfind('/');
    return k == std::string::npos ? "." : p.substr(0, k);
}

// ---- reference genome tables (src/utils/genome.rs:29-126): name -> sequences by group
struct Genome {
    std::string name;
    std::set<std::string> all, primary;
};

bool load_genome(const std::string &want, Genome *g, std::string *supported) {
    std::vector<std::string> dirs;
    if (const char *e = getenv("NGSQ_DATA_DIR")) dirs.push_back(e);
    dirs.push_back(exe_dir() + "/data");
    // the default feature set of the reference ships one genome (Cargo.toml:48-50)
    static const char *const known[] = {"GRCh38_no_alt_AnalysisSet"};
    for (const char *k : known) {
        if (!supported->empty()) *supported += ", ";
        *supported += k;
        if (!ieq(want, k)) continue;
        for (const auto &d : dirs) {
            std::ifstream f(d + "/" + k + ".tsv");
            if (!f) continue;
            g->name = k;
            std::string line;
            while (std::getline(f, line)) {
                if (line.empty() || line[0] == '#') continue;
                const size_t tab = line.find('\t');
                if (tab == std::string::npos) continue;
                const std::string nm = line.substr(0, tab), grp = line.substr(tab + 1);
                g->all.insert(nm);
                // get_primary_assembly, genome.rs:59-83: autosomes + sex + alt + unlocalized + unplaced
                if (grp == "autosome" || grp == "sex" || grp == "alt" || grp == "unlocalized" || grp == "unplaced")
                    g->primary.insert(nm);
            }
            return true;
        }
        bail(std::string("genome table ") + k + ".tsv not found (set NGSQ_DATA_DIR)");
    }
    return false;
}

// a record's number of CIGAR operations: the 16-bit column saturates at 65535, the offsets then hold the count (include/ngsq.h)
static inline uint64_t n_ops_of(const ngsq_batch &b, uint64_t i) {
    return b.cigar_off ? b.cigar_off[i + 1] - b.cigar_off[i] : b.n_cigar[i];
}

// ---- record subsets for the `-n` rules -----------------------------------------------------
// Records picked one by one (the two -n rules: command.rs:305-316 with display.rs:58-63, command.rs:354,384-388) go through the
// per-record side of the boundary, include/ngsq_stage.h -- the calls a host that keeps the reference's loops would make:
// push per record, flush when full and at the end of the pass.
struct Picker {
    ngsq_stager *s = nullptr;
    ngsq_ctx *ctx;
    uint32_t pass;
    uint64_t expect;
    Picker(ngsq_ctx *c, uint32_t pass_mask, uint64_t expect_) : ctx(c), pass(pass_mask), expect(expect_) {}
    ~Picker() { ngsq_stager_destroy(s); }
    void push(const ngsq_batch &b, uint64_t i) {
        if (!s && ngsq_stager_create(std::min<uint64_t>(std::max<uint64_t>(expect, 1), 1u << 20), NGSQ_STAGE_PINNED, &s) != NGSQ_OK)
            bail(ngsq_stager_last_error(nullptr)); // (created with the first record: -n 1000 pins a few hundred KB)
        // (the record keeps its identity -- its virtual offset from the reader, else its ordinal in the file: the GC window)
        uint64_t took = 0;
        if (ngsq_stager_push_records(s, &b, i, 1, &took) != NGSQ_OK || took != 1) bail(ngsq_stager_last_error(s));
        if (ngsq_stager_len(s) == ngsq_stager_capacity(s)) flush();
    }
    void flush() {
        if (s && ngsq_stager_flush(s, ctx, pass) != NGSQ_OK) bail(ngsq_stager_last_error(s));
    }
};

// noodles bam::Reader::query over Region(name, 1..=L) (command.rs:369-373), as in the kernels
bool query_yields(const ngsq_batch &b, uint64_t i, const std::vector<uint32_t> &ref_len) {
    const int32_t r = b.ref_id[i], p = b.pos[i];
    if (r < 0 || (size_t)r >= ref_len.size() || p < 0) return false;
    uint64_t span = 0;
    const uint32_t *c = b.cigar + (b.cigar_off ? b.cigar_off[i] : i * (uint64_t)b.cigar_stride);
    for (uint64_t k = 0, n_ops = n_ops_of(b, i); k < n_ops; k++) {
        const uint32_t op = c[k] & 0xF;
        if (op <= 8 && ((0x18Du >> op) & 1u)) span += c[k] >> 4;
    }
    const uint64_t s = (uint64_t)p + 1, e = s + span - 1;
    return e != 0 && s <= ref_len[r];
}

// (the gene model of the Genomic Features facet: gff_loader.h)
// utils/formats.rs:117-186  BioinformaticsFileFormat::try_detect, with the names its Display prints (:79-101).
// "" = no format (the callers then report the extension).  `.gz` / `.bgz` look at the whole name, case-sensitively,
// the other extensions are matched case-insensitively -- as the reference does.
std::string detect_format(const std::string &path) {
    auto ends_with = [&](const char *suf) {
        const size_t n = strlen(suf);
        return path.size() >= n && path.compare(path.size() - n, n, suf) == 0;
    };
    const size_t slash = path.rfind('/');
    const size_t dot = path.rfind('.');
    if (dot == std::string::npos || (slash != std::string::npos && dot < slash) || dot + 1 == path.size() ||
        dot == (slash == std::string::npos ? 0 : slash + 1))
        return ""; // no extension (a leading dot is not one)
    std::string ext = path.substr(dot + 1);
    for (auto &c : ext) c = (char)tolower((unsigned char)c);
    if (ext == "bgz") {
        if (ends_with("gff.bgz") || ends_with("gff3.bgz")) return "Block-gzipped GFF";
    } else if (ext == "gz") {
        if (ends_with("fasta.gz") || ends_with("fna.gz") || ends_with("fa.gz")) return "Gzipped FASTA";
        if (ends_with("fq.gz") || ends_with("fastq.gz")) return "Gzipped FASTQ";
        if (ends_with("vcf.gz")) return "Gzipped VCF";
        if (ends_with("gff.gz") || ends_with("gff3.gz")) return "Gzipped GFF";
        if (ends_with("gtf.gz")) return "Gzipped GTF";
        return "";
    }
    static const struct { const char *ext, *name; } table[] = {
        {"fasta", "FASTA"}, {"fna", "FASTA"}, {"fa", "FASTA"}, {"fastq", "FASTQ"}, {"fq", "FASTQ"}, {"sam", "SAM"},
        {"ubam", "Unaligned BAM"}, {"bam", "BAM"}, {"cram", "CRAM"}, {"vcf", "VCF"}, {"bcf", "BCF"}, {"gff", "GFF"},
        {"gff3", "GFF"}, {"gtf", "GTF"}, {"bed", "BED"}};
    for (const auto &t : table)
        if (ext == t.ext) return t.name;
    return "";
}

struct Args {
    std::string src, genome, gff, fasta, out_dir, prefix, only, vaf;
    // command.rs:77-101: GENCODE feature names by default; order = NGSQ_ROLE_*
    std::string feature_name[5] = {"five_prime_UTR", "three_prime_UTR", "CDS", "exon", "gene"};
    bool has_n = false, has_out_dir = false, has_prefix = false, has_only = false;
    unsigned long long n = 0;
    int device = 0, threads = 0;
    int coverage = 0; // --coverage auto|stream|array: 0 auto (stream when @HD says SO:coordinate), 1 stream, 2 array
    bool ingest_device = true; // --ingest host|device: where BGZF inflate + BAM parse run (ngsq_bam_next_batch[_device])
    unsigned long long batch_records = 1ull << 21, gc_seed = 0x4E4753;
    int gpus = 1;             // --gpus N: one worker process per GPU (devices --device .. --device + N - 1)
    bool same_device = false; // --same-device: all workers on --device, exchange through shared memory (one-GPU boxes)
    std::string transport;    // --transport rccl|shm: override (tests: --same-device --transport rccl with NGSQ_RCCL_LIB = tests/rccl_double)
    int rank = -1, world = 0; // --worker R/W:NAME (set by the launching process)
    std::string shm;
};

void usage() {
    fprintf(stderr,
            "Usage: ngs [-q|-v] qc [OPTIONS] <BAM> <REFERENCE_GENOME>\n\n"
            "Options:\n"
            "  -f, --features-gff <PATH>       Features GFF file (enables the Genomic Features facet)\n"
            "  -n, --num-records <USIZE>       Number of records to process in the first pass; also caps the\n"
            "                                  records per sequence in the second pass\n"
            "  -o, --output-directory <PATH>   Directory to output files to [default: current directory]\n"
            "  -p, --output-prefix <STRING>    Output prefix [default: name of the BAM file]\n"
            "  -r, --reference-fasta <PATH>    Reference FASTA file (enables the Edits facet)\n"
            "      --only <FACET>              Only process one QC facet\n"
            "      --vaf-file <PATH>           Write the VAF of every covered position (Edits facet, needs -r)\n"
            "      --five-prime-utr-feature-name <STRING>    GFF feature of a five prime UTR [default: five_prime_UTR]\n"
            "      --three-prime-utr-feature-name <STRING>   GFF feature of a three prime UTR [default: three_prime_UTR]\n"
            "      --coding-sequence-feature-name <STRING>   GFF feature of a coding sequence [default: CDS]\n"
            "      --exon-feature-name <STRING>              GFF feature of an exon [default: exon]\n"
            "      --gene-feature-name <STRING>              GFF feature of a gene [default: gene]\n"
            "      --device <N> --batch-records <N> --threads <N> --gc-seed <N> --ingest host|device   (additive, this build)\n"
            "      --coverage auto|stream|array   Coverage finished while sorted records stream by / on depth arrays (additive)\n"
            "      --gpus <N>                  One worker per GPU over BGZF block ranges of the file, one RCCL exchange (additive)\n");
}

#define CHECK(ctx, expr)                                                                                   \
    do {                                                                                                   \
        const int rc_ = (expr);                                                                            \
        if (rc_ != NGSQ_OK) bail(std::string(ngsq_last_error(ctx) && *ngsq_last_error(ctx) ? ngsq_last_error(ctx) : ngsq_last_global_error())); \
    } while (0)

} // namespace

// NGSQ_INGEST_TRACE=1: wall clock of the command's stages on stderr (measurement aid)
static void milestone(const char *what) {
    static const bool on = getenv("NGSQ_INGEST_TRACE") && atoi(getenv("NGSQ_INGEST_TRACE"));
    static const auto t0 = std::chrono::steady_clock::now();
    if (on) fprintf(stderr, "[ngs] %8.1f ms  %s\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), what);
}

int main(int argc, char **argv) {
    milestone("main");
    Args a;
    std::vector<std::string> pos;
    bool saw_qc = false;
    for (int i = 1; i < argc; i++) {
        const std::string s = argv[i];
        auto val = [&](const char *name) -> std::string {
            if (i + 1 >= argc) bail(std::string("a value is required for '") + name + "' but none was supplied");
            return argv[++i];
        };
        if (s == "-q" || s == "--quiet") g_level = 0;
        else if (s == "-v" || s == "--verbose") g_level = 3;
        else if (s == "-h" || s == "--help") { usage(); return 0; }
        else if (!saw_qc && s == "qc") saw_qc = true;
        else if (s == "-f" || s == "--features-gff") a.gff = val("--features-gff");
        else if (s == "-n" || s == "--num-records") { a.n = strtoull(val("--num-records").c_str(), nullptr, 10); a.has_n = true; }
        else if (s == "-o" || s == "--output-directory") { a.out_dir = val("--output-directory"); a.has_out_dir = true; }
        else if (s == "-p" || s == "--output-prefix") { a.prefix = val("--output-prefix"); a.has_prefix = true; }
        else if (s == "-r" || s == "--reference-fasta") a.fasta = val("--reference-fasta");
        else if (s == "--only") { a.only = val("--only"); a.has_only = true; }
        else if (s == "--vaf-file") a.vaf = val("--vaf-file");
        else if (s == "--five-prime-utr-feature-name") a.feature_name[NGSQ_ROLE_FIVE_PRIME_UTR] = val(s.c_str());
        else if (s == "--three-prime-utr-feature-name") a.feature_name[NGSQ_ROLE_THREE_PRIME_UTR] = val(s.c_str());
        else if (s == "--coding-sequence-feature-name") a.feature_name[NGSQ_ROLE_CODING_SEQUENCE] = val(s.c_str());
        else if (s == "--exon-feature-name") a.feature_name[NGSQ_ROLE_EXON] = val(s.c_str());
        else if (s == "--gene-feature-name") a.feature_name[NGSQ_ROLE_GENE] = val(s.c_str());
        else if (s == "--device") a.device = atoi(val("--device").c_str());
        else if (s == "--threads") a.threads = atoi(val("--threads").c_str());
        else if (s == "--ingest") {
            const std::string v = val("--ingest");
            if (v != "host" && v != "device") {
                fprintf(stderr, "error: --ingest takes 'host' or 'device'\n");
                return 2;
            }
            a.ingest_device = v == "device";
        }
        else if (s == "--coverage") {
            const std::string v = val("--coverage");
            if (v != "auto" && v != "stream" && v != "array") {
                fprintf(stderr, "error: --coverage takes 'auto', 'stream' or 'array'\n");
                return 2;
            }
            a.coverage = v == "auto" ? 0 : v == "stream" ? 1 : 2;
        }
        else if (s == "--gpus") a.gpus = atoi(val("--gpus").c_str());
        else if (s == "--same-device") a.same_device = true;
        else if (s == "--transport") a.transport = val("--transport");
        else if (s == "--worker") { // R/W:NAME, appended by the launching process
            const std::string v = val("--worker");
            const size_t sl = v.find('/'), co = v.find(':');
            if (sl == std::string::npos || co == std::string::npos || co < sl) bail("malformed --worker");
            a.rank = atoi(v.substr(0, sl).c_str());
            a.world = atoi(v.substr(sl + 1, co - sl - 1).c_str());
            a.shm = v.substr(co + 1);
        }
        else if (s == "--batch-records") a.batch_records = strtoull(val("--batch-records").c_str(), nullptr, 10);
        else if (s == "--gc-seed") a.gc_seed = strtoull(val("--gc-seed").c_str(), nullptr, 0);
        else if (!s.empty() && s[0] == '-') bail("unexpected argument '" + s + "' found");
        else pos.push_back(s);
    }
    if (!saw_qc) {
        usage();
        bail("this build provides the `qc` subcommand only");
    }
    if (pos.size() != 2) {
        usage();
        bail("the following required arguments were not provided: <BAM> <REFERENCE_GENOME>");
    }
    a.src = pos[0];
    a.genome = pos[1];

    // ---- qc(): command.rs:109-218
    logf(2, "Starting qc command...");
    logf(3, "  [*] Source: %s", a.src.c_str());
    Genome genome;
    std::string supported;
    if (!load_genome(a.genome, &genome, &supported))
        bail("reference genome is not supported: " + a.genome +
             ". Did you set the correct reference genome?. Use the `list genomes` subcommand to see supported reference genomes.");
    logf(3, "  [*] Reference genome: %s", a.genome.c_str());
    if (!a.has_prefix) { // default: the file name of the BAM (command.rs:156-162)
        const size_t k = a.src.rfind('/');
        a.prefix = k == std::string::npos ? a.src : a.src.substr(k + 1);
    }
    if (!a.has_out_dir) {
        char cwd[4096];
        a.out_dir = getcwd(cwd, sizeof cwd) ? cwd : ".";
    }

    // ---- app(): command.rs:226-421
    // open_and_parse(IndexCheck::Full): extension sniff, <bam>.bai must parse, header + references
    {
        const size_t dot = a.src.rfind('.');
        const std::string ext = dot == std::string::npos ? "" : a.src.substr(dot + 1);
        const std::string format = detect_format(a.src); // utils/formats/bam.rs:32-56
        if (format.empty()) bail("Not able to determine filetype for extension: " + ext);
        if (format != "BAM") bail("incompatible formats: required BAM, found " + format);
    }
    ngsq_bam *bam = nullptr;
    if (ngsq_bam_open(a.src.c_str(), a.threads, &bam) != NGSQ_OK) bail(ngsq_bam_last_error());
    milestone("header read");
    if (ngsq_bam_check_index(a.src.c_str()) != NGSQ_OK) bail(ngsq_bam_last_error());
    {
        std::error_code ec;
        std::filesystem::create_directories(a.out_dir, ec); // command.rs:186-190
        if (ec || !std::filesystem::is_directory(a.out_dir)) bail("Could not create output directory.");
    }
    const bool worker = a.world > 1;
    if (a.gpus < 1 || a.gpus > NGSQ_COMM_MAX_WORLD) bail("--gpus takes a number between 1 and 64");
    if (a.gpus > 1 && !worker && a.has_n) {
        // both truncation rules of -n are sequential and bounded by n (the first n records of the file; one counter over all
        // sequences): one process applies them, as the reference does -- N - 1 workers would only initialise their devices
        // and RCCL to bring an empty state to the exchange (and time out if rank 0 needs longer than the collective's limit)
        logf(1, "-n bounds the scan to its first records: --gpus %d is ignored, one process reads them", a.gpus);
        a.gpus = 1;
    }
    // NGSQ_RETURN_WHEN_DONE=1, one process: the command returns when the document is on disk, not when the driver has finished
    // taking the process apart (0.13-0.21 s for the 17-40 GB of device memory and the pinned buffers of a whole-genome run: a
    // quarter of the command).  The scan runs in a child forked HERE -- nothing has touched HIP yet -- which says "done" through a
    // pipe behind its last file; the parent then leaves with 0 and the child's teardown goes on behind it (a caller that starts
    // another GPU job at once finds that memory still in use for those 0.2 s, and an exit status of the teardown itself is lost:
    // hence opt-in, as for --gpus N below).  A child that ends before it has reported hands its status on.
    int single_done_fd = -1;
    if (a.gpus == 1 && !worker) {
        const char *early = getenv("NGSQ_RETURN_WHEN_DONE");
        int fds[2];
        if (early && atoi(early) && pipe(fds) == 0) {
            fflush(nullptr);
            const pid_t child = fork();
            if (child > 0) {
                close(fds[1]);
                char b;
                const ssize_t n = read(fds[0], &b, 1); // 1: done; 0: the child has ended without saying so
                if (n == 1) _exit(0);
                int status = 0;
                waitpid(child, &status, 0);
                _exit(WIFEXITED(status) ? WEXITSTATUS(status) : 1);
            }
            if (child == 0) {
                close(fds[0]);
                single_done_fd = fds[1];
            } else { // (no fork: carry on as one process)
                close(fds[0]);
                close(fds[1]);
            }
        }
    }
    if (a.gpus > 1 && !worker) {
        // ---- launch one worker per GPU.  Nothing above has touched HIP, and nothing here does: the workers
        // are fresh processes (posix_spawn of this executable), each initialises its own device.
        if (!a.ingest_device) bail("--gpus needs --ingest device");
        ngsq_bam_close(bam);
        char shm[128];
        snprintf(shm, sizeof shm, "/ngsq-cli-%d-%lld", (int)getpid(), (long long)time(nullptr));
        // The command returns when every worker has EXITED and hands on the worst of their exit statuses (round 4).
        // NGSQ_RETURN_WHEN_DONE=1 is the opt-in for callers that only want the document: a worker then says "done" through
        // this pipe when the document is on disk and nothing of it is left to do, and the command returns when all of them
        // have -- while the kernel is still unmapping their GiB of device and pinned memory (0.1-0.35 s, serialised among
        // the workers of one device): such a caller inherits that memory still in use, and loses the exit statuses.
        int done_fd[2] = {-1, -1};
        const char *early = getenv("NGSQ_RETURN_WHEN_DONE");
        if (early && atoi(early) && pipe(done_fd) == 0) {
            setenv("NGSQ_DONE_FD", std::to_string(done_fd[1]).c_str(), 1);
        } else {
            done_fd[0] = done_fd[1] = -1;
            unsetenv("NGSQ_DONE_FD"); // (a stale value must not reach the workers)
        }
        // the host driver of these machines supports dmabuf IPC only: without this RCCL's peer-memory set-up fails with
        // "hipIpcGetMemHandle: invalid argument" (bench.py's launcher sets the same default for its ranks)
        setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);
        // fewer than six cores per worker (four pread threads, the reader, the driver: DESIGN.md section 8): the thread
        // that waits for the GPU sleeps instead of spinning on a core the reader's copies need
        if (cgroup_cores() < 6 * a.gpus) setenv("NGSQ_BLOCKING_SYNC", "1", 0);
        std::vector<pid_t> pids;
        for (int r = 0; r < a.gpus; r++) {
            std::vector<char *> av(argv, argv + argc);
            char opt[] = "--worker";
            std::string spec = std::to_string(r) + "/" + std::to_string(a.gpus) + ":" + shm;
            av.push_back(opt);
            av.push_back(spec.data());
            av.push_back(nullptr);
            pid_t pid;
            if (posix_spawn(&pid, "/proc/self/exe", nullptr, nullptr, av.data(), environ) != 0) bail("could not start a worker process");
            pids.push_back(pid);
        }
        milestone("workers started");
        if (done_fd[1] >= 0) close(done_fd[1]);
        int worst = 0;
        size_t done = 0;
        for (size_t left = pids.size(); left;) {
            if (done_fd[0] >= 0) {
                struct pollfd pf = {done_fd[0], POLLIN, 0};
                if (poll(&pf, 1, 2) > 0 && (pf.revents & POLLIN)) {
                    char buf[64];
                    const ssize_t n = read(done_fd[0], buf, sizeof buf);
                    if (n > 0) done += (size_t)n;
                }
                if (done >= pids.size() && !worst) {
                    milestone("every worker has reported");
                    return 0;
                }
            }
            int status = 0;
            const pid_t p = waitpid(-1, &status, done_fd[0] >= 0 ? WNOHANG : 0);
            if (p == 0) continue;
            if (p < 0) break;
            left--;
            const int code = WIFEXITED(status) ? WEXITSTATUS(status) : 1;
            if (code && !worst) { // a failed worker leaves the others waiting in a collective: stop exactly those
                worst = code;
                for (pid_t q : pids)
                    if (q != p) kill(q, SIGTERM);
            }
            milestone("a worker has exited");
        }
        return worst;
    }
    if (worker && a.rank > 0) g_level = std::min(g_level, 1); // rank 0 narrates
    const uint32_t n_refs = ngsq_bam_n_refs(bam);
    std::vector<std::string> names(n_refs);
    std::vector<uint32_t> ref_len(n_refs);
    std::vector<uint8_t> primary(n_refs);
    for (uint32_t r = 0; r < n_refs; r++) {
        names[r] = ngsq_bam_ref_name(bam, r);
        ref_len[r] = ngsq_bam_ref_len(bam, r);
        if (!genome.all.count(names[r])) // command.rs:258-272
            bail("Sequence \"" + names[r] + "\" not found in specified reference genome. Did you set the correct reference genome?");
        primary[r] = genome.primary.count(names[r]) ? 1 : 0; // coverage.rs:133-138
    }

    // facets: qc.rs:44-126
    uint32_t facets = NGSQ_FACET_GENERAL | NGSQ_FACET_TEMPLATE_LENGTH | NGSQ_FACET_GC_CONTENT | NGSQ_FACET_QUALITY_SCORE |
                      NGSQ_FACET_COVERAGE;
    if (!a.fasta.empty()) facets |= NGSQ_FACET_EDITS;
    // Genomic Features: the gene model is read while the facets are built (qc.rs:68-79), before --only.  Here it is read by
    // threads of its own (gff_loader.h) WHILE the device is initialised below; its errors are reported where the reference
    // reports them -- before anything of the scan happens.
    GeneModel model;
    std::thread gff_thread;
    std::atomic<bool> gff_done{false};
    if (!a.gff.empty()) {
        const size_t dot = a.gff.rfind('.');
        const std::string ext = dot == std::string::npos ? "" : a.gff.substr(dot + 1);
        const std::string format = detect_format(a.gff); // utils/formats/gff.rs:19-47
        if (format.empty()) bail("opening GFF file: " + a.gff + ": Not able to determine filetype for extension: " + ext);
        if (format != "GFF" && format != "Gzipped GFF") bail("opening GFF file: " + a.gff + ": incompatible formats: required GFF, found " + format);
        {
            struct stat st;
            if (stat(a.gff.c_str(), &st) != 0) bail("opening GFF file: " + a.gff + ": No such file or directory (os error 2)");
        }
        logf(3, "Reading all records in GFF.");
        const bool gz = format == "Gzipped GFF";
        const int nt = std::max(1, std::min(8, cgroup_cores() - 2));
        gff_thread = std::thread([&model, &a, &genome, &names, &gff_done, n_refs, gz, nt] {
            std::map<std::string, uint32_t> ref_index;
            for (uint32_t r = 0; r < n_refs; r++) ref_index[names[r]] = r;
            model = load_gff_parallel(a.gff, gz, a.feature_name, genome.primary, ref_index, nt);
            gff_done = true;
        });
        facets |= NGSQ_FACET_FEATURES;
    }
    auto gff_ready = [&]() { // (at most once)
        if (!gff_thread.joinable()) return;
        gff_thread.join();
        if (!model.error.empty()) bail(model.error);
        logf(3, "Tabulating GFF features.");
        logf(3, "Finalizing GFF features lookup.");
        if (getenv("NGSQ_INGEST_TRACE") && atoi(getenv("NGSQ_INGEST_TRACE")))
            fprintf(stderr, "[ngs] gene model: %.1f MB of GFF, %llu lines, %zu intervals kept, %.1f ms on its threads\n", model.text_bytes / 1e6,
                    (unsigned long long)model.lines, model.ref.size(), model.seconds * 1e3);
    };
    // EditsFacet::try_from (edits.rs:120-151): the FASTA is opened once "to make sure that all is well" -- formats::fasta::open
    // (utils/formats/fasta.rs:15-41) decides by the extension -- and the VAF file is created, both before --only filters the
    // facets.  Opening it HERE also starts the index of its definition lines on threads of the library (include/ngsq_reference.h),
    // beside the gene model and the device's initialisation.
    ngsq_fasta *fasta = nullptr;
    if (!a.fasta.empty()) {
        const size_t dot = a.fasta.rfind('.');
        const std::string ext = dot == std::string::npos ? "" : a.fasta.substr(dot + 1);
        const std::string format = detect_format(a.fasta);
        const std::string ctxt = "opening reference FASTA file: " + a.fasta + ": ";
        if (format == "Gzipped FASTA") bail(ctxt + "This command does not yet support gzipped FASTA files. Please unzip your FASTA file and try again.");
        if (format.empty()) bail(ctxt + "Not able to determine filetype for extension: " + ext);
        if (format != "FASTA") bail(ctxt + "incompatible formats: required FASTA, found " + format);
        // (a sharded run's workers share the cores: a FASTA thread or two each)
        if (ngsq_fasta_open(a.fasta.c_str(), worker ? std::max(1, std::min(4, cgroup_cores() / std::max(1, a.world) - 1)) : 0, &fasta) != NGSQ_OK)
            bail(ctxt + ngsq_fasta_last_error());
    }
    const bool want_vaf = !a.fasta.empty() && !a.vaf.empty() && (a.world <= 1 || a.rank == 0); // one writer in a --gpus run
    if (want_vaf) {
        struct stat st;
        if (stat(a.vaf.c_str(), &st) == 0)
            bail("refusing to overwrite existing VAF file: " + a.vaf + ". Please delete and rerun if you'd like to replace it.");
    }
    if (a.has_only) {
        uint32_t sel = 0;
        int matched = 0;
        for (uint32_t bit = 1; bit <= NGSQ_FACET_FEATURES; bit <<= 1)
            if ((facets & bit) && ieq(a.only, ngsq_facet_name(bit))) {
                sel |= bit;
                matched++;
            }
        if (matched == 0) {
            gff_ready(); // (the reference has read the GFF by now: its errors come first)
            bail("No facets matched the specified `--only` flag: " + a.only);
        }
        facets = sel;
    }
    FILE *vaf_file = nullptr;
    // which sequences' bases this process needs: all of them -- or, for a worker of a --gpus run, the sequences whose records
    // its byte range of the (sorted, indexed) file can hold; the others are never uploaded (N workers do not read the FASTA N times)
    std::vector<uint8_t> ref_wanted;
    if ((facets & NGSQ_FACET_EDITS) && worker && !a.has_n) {
        std::vector<uint64_t> ref_start(n_refs, 0);
        uint64_t index_bins = 0;
        struct stat st;
        if (ngsq_bam_index_ref_starts(a.src.c_str(), n_refs, ref_start.data(), &index_bins) == NGSQ_OK && index_bins > 0 && stat(a.src.c_str(), &st) == 0) {
            // (a record belongs to the worker its first byte lies in; the margin covers the rounding of a range's end to a block start --
            // and is generous: a sequence too many costs 0.1 s, one too few fails the run.  NGSQ_REF_MARGIN_BYTES: tests)
            const uint64_t size = (uint64_t)st.st_size, margin = getenv("NGSQ_REF_MARGIN_BYTES") ? strtoull(getenv("NGSQ_REF_MARGIN_BYTES"), nullptr, 10) : (uint64_t)20 << 20;
            const uint64_t lo = size / (uint64_t)a.world * (uint64_t)a.rank, hi = a.rank + 1 == a.world ? size : size / (uint64_t)a.world * (uint64_t)(a.rank + 1) + margin;
            ref_wanted.assign(n_refs, 0);
            int64_t prev = -1; // the last sequence in front with records
            for (uint32_t r = 0; r < n_refs; r++) {
                if (!ref_start[r]) continue;
                const uint64_t begin = ref_start[r] >> 16;
                uint64_t end = size;
                for (uint32_t q = r + 1; q < n_refs; q++)
                    if (ref_start[q]) {
                        end = (ref_start[q] >> 16) + 65536; // (the block the next sequence starts in may still hold this one's records)
                        break;
                    }
                if (begin < hi && end > lo) {
                    ref_wanted[r] = 1;
                    if (prev >= 0) ref_wanted[(size_t)prev] = 1; // (and its neighbour in front, for good measure)
                }
                prev = r;
            }
        }
    }

    ngsq_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = sizeof cfg;
    cfg.facets = facets;
    cfg.device = a.device;
    cfg.n_refs = n_refs;
    cfg.ref_len = ref_len.data();
    cfg.ref_is_primary = primary.data();
    cfg.bin_size = 50000;  // qc.rs:87
    cfg.tlen_cap = 1024;   // qc.rs:62
    cfg.cov_cap = 2048;    // coverage.rs:76
    cfg.max_read_len = 256; // where the quality table starts: it grows with the longest read of the file
    cfg.gc_seed = a.gc_seed;
    cfg.ref_bases = nullptr;
    cfg.ref_bases_deferred = (facets & NGSQ_FACET_EDITS) ? 1 : 0; // the bases come from the file: ngsq_reference_load below
    // Coverage while the records stream by (ngsq_config.sorted_input) needs coordinate order.  `ngs qc` only
    // accepts indexed, i.e. sorted, files (formats/bam.rs:86-96); "auto" takes the header's word for it and
    // falls back to the depth arrays when a record turns out to break the order.
    bool header_sorted = false;
    {
        uint64_t hl = 0;
        const char *ht = ngsq_bam_header_text(bam, &hl);
        const std::string text(ht ? ht : "", ht ? hl : 0);
        const size_t hd = text.rfind("@HD", 0) == 0 ? 0 : text.find("\n@HD");
        if (hd != std::string::npos) {
            const size_t eol = text.find('\n', hd + 1);
            header_sorted = text.substr(hd, eol == std::string::npos ? std::string::npos : eol - hd).find("SO:coordinate") != std::string::npos;
        }
    }
    // ---- the communicator of a --gpus run: the workers meet in the shared-memory segment the launcher named;
    // with one device per worker rank 0's RCCL unique id travels through it and the exchange runs over xGMI
    ngsq_comm *comm = nullptr;
    std::thread rccl_init;          // RCCL's initialisation, beside the scan (below)
    ngsq_comm *rccl_comm = nullptr; // its result ...
    std::string rccl_error;         // ... or why there is none
    if (worker) {
        const int ndev = ngsq_device_count();
        if (ndev < 1) bail("no HIP device available; the ngs qc hot path has no CPU fallback");
        ngsq_comm *boot = nullptr;
        if (ngsq_comm_create_shm(a.shm.c_str(), a.rank, a.world, 0, &boot) != NGSQ_OK) bail(ngsq_comm_last_error(nullptr));
        // (auto = what a run on a device per worker does without the option -- RCCL if it comes up, else shared memory -- also
        // with --same-device, where RCCL refuses to come up: the fallback's test)
        if (!a.transport.empty() && a.transport != "rccl" && a.transport != "shm" && a.transport != "auto") bail("--transport must be rccl, shm or auto");
        const bool rccl_named = a.transport == "rccl";
        if (a.transport.empty() ? a.same_device : a.transport == "shm") {
            if (!a.same_device) a.device += a.rank;
            comm = boot;
        } else {
            if (!a.same_device) {
                if (a.device + a.world > ndev)
                    bail("--gpus " + std::to_string(a.world) + " from --device " + std::to_string(a.device) + ": only " + std::to_string(ndev) +
                         " device(s) visible (--same-device shares one)");
                a.device += a.rank;
            }
            // rank 0's unique id + "I have one": a rank 0 that cannot load librccl must not leave the others waiting
            uint8_t uid[NGSQ_COMM_ID_BYTES + 8] = {0};
            std::string no_rccl;
            if (a.rank == 0) {
                if (ngsq_comm_unique_id(uid) == NGSQ_OK) uid[NGSQ_COMM_ID_BYTES] = 1;
                else no_rccl = ngsq_comm_last_error(nullptr);
            }
            std::vector<uint8_t> all((size_t)a.world * sizeof uid);
            if (ngsq_comm_allgather_host(boot, uid, all.data(), sizeof uid) != NGSQ_OK) bail(ngsq_comm_last_error(boot));
            if (!all[NGSQ_COMM_ID_BYTES]) {
                if (rccl_named) bail(a.rank == 0 ? no_rccl : std::string("RCCL is not available on rank 0"));
                // not asked for by name: the same exchange, host-staged through the shared-memory segment
                if (a.rank == 0) logf(1, "%s; the exchange runs over shared memory instead", no_rccl.c_str());
                comm = boot;
            } else {
                // ncclCommInitRank (bootstrap over sockets, topology search, channel set-up: the better part of a second on a
                // node of eight GPUs, several times the scan of this worker's share of a file) runs on its own thread beside
                // the scan: the scan asks the communicator for its rank and size only (the shared-memory one answers that),
                // the first message is the boundary check behind it (comm_ready below).
                comm = boot;
                const int rank = a.rank, world = a.world, device = a.device;
                rccl_init = std::thread([&rccl_comm, &rccl_error, all, rank, world, device]() {
                    if (ngsq_comm_create_rccl(rank, world, all.data(), device, &rccl_comm) != NGSQ_OK) {
                        rccl_error = ngsq_comm_last_error(nullptr);
                        if (rccl_error.empty()) rccl_error = "RCCL's communicator could not be created";
                        rccl_comm = nullptr;
                    }
                });
            }
        }
        cfg.device = a.device;
        logf(2, "Worker %d of %d on device %d, exchange over %s.", a.rank, a.world, a.device, rccl_init.joinable() ? "rccl" : ngsq_comm_kind(comm));
    }
    // before the first message of a --gpus run: RCCL's communicator takes the place of the one the workers met in.  Every
    // rank says whether it has one: a rank whose initialisation failed must not leave the others waiting inside RCCL.
    auto comm_ready = [&]() {
        if (!rccl_init.joinable()) return;
        rccl_init.join(); // (bounded: ngsq_comm_create_rccl gives up after NGSQ_RCCL_INIT_TIMEOUT_S)
        // RCCL was asked for by name and this worker has no communicator: leave NOW -- the launcher stops the others, which
        // may be inside ncclCommInitRank waiting for this one (a vote first would wait for them: ADVICE r3)
        if (!rccl_comm && a.transport == "rccl") bail(rccl_error);
        ngsq_comm *boot = comm;
        const uint8_t ok = rccl_comm != nullptr;
        std::vector<uint8_t> oks((size_t)a.world);
        if (ngsq_comm_allgather_host(boot, &ok, oks.data(), 1) != NGSQ_OK) bail(ngsq_comm_last_error(boot));
        for (int r = 0; r < a.world; r++)
            if (!oks[(size_t)r]) {
                const std::string why = r == a.rank ? rccl_error : "RCCL's communicator could not be created on worker " + std::to_string(r);
                if (a.transport == "rccl") bail(why);
                // not asked for by name: the same exchange, host-staged through the shared-memory segment the workers met in
                // (a communicator some ranks did get is left to the end of the process: destroying half of one may not return)
                if (a.rank == 0) logf(1, "%s; the exchange runs over shared memory instead", why.c_str());
                return;
            }
        comm = rccl_comm;
        if (ngsq_comm_barrier(boot) != NGSQ_OK) bail(ngsq_comm_last_error(boot));
        ngsq_comm_destroy(boot);
    };
    ngsq_ctx *ctx = nullptr;
    unsigned long long n_pass1 = 0;
    for (bool force_array = false;;) { // at most twice: again on the depth arrays when the records break the promised order
    cfg.sorted_input = (!force_array && !a.has_n && (facets & NGSQ_FACET_COVERAGE) && (a.coverage == 1 || (a.coverage == 0 && header_sorted))) ? 1 : 0;
    // shards behind the first: a read of the shard in front may reach this far into this shard's first positions
    cfg.cov_head_guard = (worker && a.rank > 0 && cfg.sorted_input) ? (1u << 20) : 0;
    milestone("checks done");
    if (ngsq_create(&cfg, &ctx) != NGSQ_OK) bail(ngsq_last_global_error());
    milestone("context created (HIP initialised)");
    auto start_reference = [&]() {
        if (!(facets & NGSQ_FACET_EDITS)) return;
        // EditsFacet::setup for every sequence of the header (edits.rs:177-215), as one pass over the file: the text goes to the
        // device as it is, on the library's threads, while this thread goes on to the first batches (ngsq_process_batch waits
        // for it in front of the first Edits kernel only)
        std::vector<const char *> name_ptrs(n_refs ? n_refs : 1, "");
        for (uint32_t r = 0; r < n_refs; r++) name_ptrs[r] = names[r].c_str();
        CHECK(ctx, ngsq_reference_load(ctx, fasta, name_ptrs.data(), ref_wanted.empty() ? nullptr : ref_wanted.data()));
        if (!ref_wanted.empty()) {
            uint32_t n_w = 0;
            for (uint8_t w : ref_wanted) n_w += w;
            logf(3, "  [*] Worker %d: the bases of %u of the %u sequences.", a.rank, n_w, n_refs);
        }
        milestone("reference load started");
    };
    auto install_gene_model = [&]() {
        gff_ready();
        milestone("gene model ready");
        if (!(facets & NGSQ_FACET_FEATURES)) return;
        ngsq_features f;
        memset(&f, 0, sizeof f);
        f.struct_size = sizeof f;
        for (int k = 0; k < 5; k++) f.role_name[k] = model.role_name[k];
        f.n = model.ref.size();
        f.ref_id = model.ref.data();
        f.name = model.name.data();
        f.start = model.start.data();
        f.stop = model.stop.data();
        CHECK(ctx, ngsq_set_features(ctx, &f));
        milestone("gene model on the device");
    };
    // The gene model is usually there by now (it was parsed beside the device's initialisation): it goes to the device FIRST -- its
    // synchronous copies took 0.4 s when they queued behind the reference's 3 GB of text (round 6) -- and the reference starts
    // behind it; a model that is still being read (a gzip stream inflates on one thread) waits beside the reference instead.
    bool model_pending = false;
    if (!gff_thread.joinable() || gff_done) {
        install_gene_model();
        start_reference();
    } else {
        // (the scan does not wait for it: batches that come before the model keep what the facet needs of their records on the
        // device -- ngsq_process_batch, 16 bytes per record -- and ngsq_set_features looks them up when it arrives: a gzipped GFF
        // inflates on one thread for seconds, beside a scan of seconds)
        start_reference();
        model_pending = (facets & NGSQ_FACET_FEATURES) != 0;
        if (!model_pending) install_gene_model();
    }
    auto model_if_ready = [&](bool wait) {
        if (model_pending && (wait || gff_done)) {
            install_gene_model();
            model_pending = false;
        }
    };
    if (want_vaf && !vaf_file) {
        vaf_file = fopen(a.vaf.c_str(), "wb");
        if (!vaf_file) bail("creating VAF file");
        fputs("Sequence\tPosition\tVAF\n", vaf_file);
    }

    const bool rec_facets = (facets & NGSQ_FACETS_RECORD_BASED) != 0, seq_facets = (facets & NGSQ_FACETS_SEQUENCE_BASED) != 0;
    if (rec_facets) {
        logf(2, "First pass with the following facets enabled:");
        static const char *load[] = {"Light", "Light", "Light", "Moderate"};
        for (int k = 0; k < 4; k++)
            if (facets & (1u << k)) logf(2, "  [*] %s, %s", ngsq_facet_name(1u << k), load[k]);
        if (facets & NGSQ_FACET_FEATURES) logf(2, "  [*] Genomic Features, Moderate"); // features.rs:107-113
        logf(2, "Starting first pass for QC stats.");
    } else {
        logf(2, "No facets specified that require first pass. Skipping...");
    }

    n_pass1 = 0;
    bool shard_unsorted = false;
    if (worker && !a.has_n) {
        // this worker's BGZF block range, streamed through the same chunked pipeline as a whole file; the shards
        // compare the record boundaries they assumed when all have reached their end (ngsq_bam_shard_verify), and
        // a shard whose assumption was wrong scans again from the confirmed offset
        if (ngsq_bam_shard_open(bam, ctx, comm) != NGSQ_OK) bail(ngsq_comm_last_error(comm));
        bool scanning = true;
        std::string scan_error; // a worker that fails still meets the others in the collective: nobody waits for it
        for (;;) {
            if (scanning) {
                scan_error.clear();
                n_pass1 = 0;
                for (;;) {
                    ngsq_batch b;
                    if (ngsq_bam_next_batch_device(bam, ctx, a.batch_records, &b) != NGSQ_OK) {
                        scan_error = ngsq_bam_last_error();
                        break;
                    }
                    if (!b.n_records) break;
                    if (ngsq_process_batch(ctx, &b, NGSQ_PASS_BOTH) != NGSQ_OK) {
                        scan_error = ngsq_last_error(ctx);
                        break;
                    }
                    model_if_ready(false);
                    n_pass1 += b.n_records;
                }
            }
            ngsq_bam_shard_info info;
            int again = 0;
            model_if_ready(true);
            comm_ready();
            const int vrc = ngsq_bam_shard_verify(bam, ctx, comm, &info, &again);
            // (a scan that failed while it ran from an ASSUMED first record is forgiven once: the verdict is `again`.  Only the
            // round that scanned reports its own failure: a worker that keeps its state while its predecessor is still being
            // re-armed must not answer a LATER round's failure -- another shard's, the transport's -- with the error it was
            // forgiven for: ADVICE r4)
            if (vrc != NGSQ_OK && scanning && !scan_error.empty()) bail(scan_error);
            if (vrc == NGSQ_ERR_UNSORTED) { // neighbouring shards out of coordinate order: same verdict on every worker
                shard_unsorted = true;
                break;
            }
            if (vrc != NGSQ_OK) bail(ngsq_comm_last_error(comm));
            if (!again) {
                logf(3, "  [*] Worker %d: %llu records, records in front of it: %llu.", a.rank, (unsigned long long)info.n_records,
                     (unsigned long long)info.first_record_index);
                break;
            }
            const std::string forgiven = scan_error;
            scan_error.clear();
            scanning = info.rescan != 0;
            if (scanning) {
                logf(1, "worker %d: the assumed first record of its shard was not one%s; scanning the shard again from the confirmed offset", a.rank,
                     forgiven.empty() ? "" : (" (" + forgiven + ")").c_str());
                CHECK(ctx, ngsq_reset(ctx));
            }
        }
    } else if (!a.has_n) {
        // no truncation: both passes see every record -> one scan
        for (;;) {
            ngsq_batch b;
            if ((a.ingest_device ? ngsq_bam_next_batch_device(bam, ctx, a.batch_records, &b)
                                 : ngsq_bam_next_batch(bam, a.batch_records, &b)) != NGSQ_OK)
                bail(ngsq_bam_last_error());
            if (!b.n_records) break;
            CHECK(ctx, ngsq_process_batch(ctx, &b, NGSQ_PASS_BOTH));
            model_if_ready(false);
            const unsigned long long before = n_pass1;
            n_pass1 += b.n_records;
            for (unsigned long long m = before / 1000000 + 1; m * 1000000 <= n_pass1; m++)
                logf(2, "  [*] Processed %s records.", with_commas(m * 1000000).c_str()); // display.rs:43-52
        }
    } else if (worker && a.rank != 0) {
        // -n with --gpus: both truncation rules are sequential and bounded by n (the first n records of the file; one
        // counter over all sequences), so worker 0 applies them exactly as a single process does -- host reader, region
        // queries through the index -- and the others bring an empty state to the exchange
    } else {
        // pass 1: stop after exactly n records (display.rs:58-63, `>=` after the increment)
        // pass 2: one counter over all sequences (command.rs:354,384-388): a sequence stops once the
        //         counter has reached n, so every later sequence still processes ONE record.
        std::vector<std::vector<unsigned long long>> yielded(n_refs); // file indices, first max(n,1) per sequence
        const unsigned long long keep = std::max<unsigned long long>(a.n, 1);
        // With a real index the sequence pass does what the reference does (command.rs:356-397): one region query
        // per sequence -- the index gives the first record of the sequence, the reader seeks there and stops at
        // the counter or at the next sequence -- and nothing else of the file is read.  An index without bins
        // (nothing to look up) falls back to one scan of the file.
        std::vector<uint64_t> ref_start(n_refs, 0);
        uint64_t index_bins = 0;
        if (seq_facets && ngsq_bam_index_ref_starts(a.src.c_str(), n_refs, ref_start.data(), &index_bins) != NGSQ_OK)
            bail(ngsq_bam_last_error());
        const bool by_index = seq_facets && index_bins > 0;
        Picker pass1(ctx, NGSQ_PASS_RECORD, keep), pass2(ctx, NGSQ_PASS_SEQUENCE, keep + n_refs);
        for (;;) {
            ngsq_batch b;
            // (when only the first `keep` records are wanted from this loop, do not decode a whole batch)
            const unsigned long long want = (!seq_facets || by_index) ? std::min<unsigned long long>(a.batch_records, keep - std::min(keep, n_pass1) + 1)
                                                                      : a.batch_records;
            if (ngsq_bam_next_batch(bam, want, &b) != NGSQ_OK) bail(ngsq_bam_last_error());
            if (!b.n_records) break;
            if (rec_facets && (n_pass1 < keep)) {
                const unsigned long long take = std::min<unsigned long long>(b.n_records, keep - n_pass1);
                for (unsigned long long i = 0; i < take; i++) pass1.push(b, i);
                n_pass1 += take;
            }
            if (seq_facets && !by_index)
                for (unsigned long long i = 0; i < b.n_records; i++)
                    if (query_yields(b, i, ref_len)) {
                        auto &v = yielded[b.ref_id[i]];
                        if (v.size() < keep) v.push_back(b.first_record_index + i);
                    }
            if ((!seq_facets || by_index) && n_pass1 >= keep) break;
            if (!rec_facets && by_index) break;
        }
        pass1.flush(); // `summarize`: the end of pass 1 (command.rs:328-330)
        if (by_index) {
            Picker &c = pass2;
            unsigned long long counter = 0, queries = 0;
            for (uint32_t r = 0; r < n_refs; r++) {
                if (!ref_start[r]) continue; // the index holds nothing for this sequence
                if (ngsq_bam_seek(bam, ref_start[r]) != NGSQ_OK) bail(ngsq_bam_last_error());
                queries += 1;
                bool done = false;
                while (!done) {
                    ngsq_batch b;
                    if (ngsq_bam_next_batch(bam, std::min<unsigned long long>(a.batch_records, 4096), &b) != NGSQ_OK)
                        bail(ngsq_bam_last_error());
                    if (!b.n_records) break;
                    for (unsigned long long i = 0; i < b.n_records && !done; i++) {
                        if (b.ref_id[i] != (int32_t)r) { // the sorted file has moved on (the chunk may begin a little early)
                            done = b.ref_id[i] > (int32_t)r || b.ref_id[i] < 0;
                            continue;
                        }
                        if (!query_yields(b, i, ref_len)) continue;
                        c.push(b, i);
                        counter += 1;
                        if (counter >= a.n) done = true; // one counter over all sequences: command.rs:354,384-388
                    }
                }
            }
            logf(3, "  [*] %llu region queries through the index.", queries);
            c.flush(); // (sequence facets do not use the record's identity)
        } else if (seq_facets) {
            std::set<unsigned long long> picks;
            unsigned long long counter = 0;
            for (uint32_t r = 0; r < n_refs; r++)
                for (unsigned long long idx : yielded[r]) {
                    picks.insert(idx);
                    counter += 1;
                    if (counter >= a.n) break;
                }
            ngsq_bam_close(bam);
            if (ngsq_bam_open(a.src.c_str(), a.threads, &bam) != NGSQ_OK) bail(ngsq_bam_last_error());
            Picker &c = pass2;
            for (;;) {
                ngsq_batch b;
                if (ngsq_bam_next_batch(bam, a.batch_records, &b) != NGSQ_OK) bail(ngsq_bam_last_error());
                if (!b.n_records) break;
                auto lo = picks.lower_bound(b.first_record_index);
                for (; lo != picks.end() && *lo < b.first_record_index + b.n_records; ++lo) c.push(b, *lo - b.first_record_index);
            }
            c.flush();
        }
    }
    if (rec_facets) {
        logf(2, "Processed %s records in the first pass.", with_commas(n_pass1).c_str());
        logf(2, "Summarizing quality control facets for the first pass.");
    }
    if (seq_facets) {
        logf(2, "Second pass with the following facets enabled:");
        if (facets & NGSQ_FACET_COVERAGE) logf(2, "  [*] Coverage, Moderate");
        if (facets & NGSQ_FACET_EDITS) logf(2, "  [*] Edits, Heavy");
        logf(2, "Starting second pass for QC stats.");
    } else {
        logf(2, "No facets specified that require second pass. Skipping...");
    }
    {
        int rc = NGSQ_OK;
        std::string why;
        model_if_ready(true); // (at the latest: the facet's tallies are part of what is exchanged and finalized)
        comm_ready();
        if (shard_unsorted) {
            rc = NGSQ_ERR_UNSORTED;
            why = ngsq_comm_last_error(comm);
        } else if (worker && (rc = ngsq_exchange(ctx, comm, nullptr)) != NGSQ_OK) why = ngsq_comm_last_error(comm);
        milestone("records scanned");
        // (a file without records never asked for the reference: its sequences are looked up all the same, edits.rs:207-209)
        if (rc == NGSQ_OK && (facets & NGSQ_FACET_EDITS) && (rc = ngsq_reference_wait(ctx)) != NGSQ_OK) why = ngsq_last_error(ctx);
        if (rc == NGSQ_OK && (facets & NGSQ_FACET_EDITS) && getenv("NGSQ_INGEST_TRACE") && atoi(getenv("NGSQ_INGEST_TRACE"))) {
            ngsq_reference_stats rs;
            if (ngsq_reference_get_stats(ctx, &rs) == NGSQ_OK)
                fprintf(stderr, "[ngs] reference: %.1f MB of FASTA text, %u sequences, %llu bases; index %s %.1f ms (waited %.1f), read + copy %.1f ms, kernels %.1f ms, load %.1f ms\n",
                        rs.text_bytes / 1e6, rs.sequences, (unsigned long long)rs.bases, ngsq_fasta_index_from_fai(fasta) ? "(.fai)" : "(scan)",
                        ngsq_fasta_index_seconds(fasta) * 1e3, rs.index_wait_s * 1e3, rs.read_s * 1e3, rs.device_s * 1e3, rs.total_s * 1e3);
        }
        if (rc == NGSQ_OK && (rc = ngsq_finalize(ctx)) != NGSQ_OK) why = ngsq_last_error(ctx);
        milestone("finalized");
        if (rc == NGSQ_ERR_UNSORTED && a.coverage == 0 && !force_array) {
            // the header promised coordinate order and the records broke it (every worker of a --gpus run sees the
            // same verdict: the counters are summed): scan again, in this process, on the depth arrays
            logf(1, "records are not in coordinate order although the header says so: scanning again with --coverage array");
            ngsq_destroy(ctx);
            ctx = nullptr;
            ngsq_bam_close(bam);
            if (ngsq_bam_open(a.src.c_str(), a.threads, &bam) != NGSQ_OK) bail(ngsq_bam_last_error());
            force_array = true;
            continue;
        }
        if (rc != NGSQ_OK) bail(why);
        break;
    }
    } // scan attempts
    // A worker process has nothing left to do once the document is on disk: by default it leaves its GiB of device and
    // pinned memory to the kernel's process teardown instead of unmapping them block by block and running the HIP
    // runtime's exit handlers (three workers on one device: 0.37 s of the command's 1.07; NGSQ_QUICK_EXIT=0 turns it off)
    auto report_done = [&]() { // (the launcher's pipe: see there)
        if (single_done_fd >= 0) {
            const char ok = 0;
            if (write(single_done_fd, &ok, 1) != 1) { /* the parent then waits for the exit status instead */ }
            return;
        }
        const char *fd = worker ? getenv("NGSQ_DONE_FD") : nullptr;
        const char ok = 0;
        if (fd && write(atoi(fd), &ok, 1) != 1) { /* the launcher then waits for the exit status instead */ }
    };
    // (with RCCL's communicator alive the worker goes through ngsq_comm_destroy first: leaving without ncclCommDestroy has
    // not been seen on a multi-GPU node yet; a thread stuck in ncclCommInitRank forces the quick way out)
    const char *qe = getenv("NGSQ_QUICK_EXIT");
    // (round 6: the single process leaves the same way -- 0.1-0.2 s of unmapping and exit handlers behind a document that is
    // already on disk were a fifth of the command's wall clock on a 6 GB file)
    const bool quick_exit = ngsq_comm_rccl_stuck() || (qe ? atoi(qe) != 0 : !worker || std::string(ngsq_comm_kind(comm)) != "rccl");
    if (worker && a.rank != 0) { // every rank holds the whole-file result; rank 0 writes it
        if (vaf_file) fclose(vaf_file);
        ngsq_comm_barrier(comm);
        if (quick_exit) {
            fflush(nullptr);
            report_done();
            _exit(0);
        }
        ngsq_destroy(ctx);
        ngsq_bam_close(bam);
        ngsq_comm_destroy(comm);
        return 0;
    }
    if (vaf_file && (facets & NGSQ_FACET_EDITS)) {
        // edits.rs:320-341, per sequence in header order: one line per position any record covered; the
        // value is the f32 the histogram bin was taken from, printed as Rust prints it (shortest digits
        // that round-trip, never an exponent)
        std::vector<uint32_t> refs, alts;
        for (uint32_t r = 0; r < n_refs; r++) {
            const size_t L1 = (size_t)ref_len[r] + 1;
            refs.resize(L1);
            alts.resize(L1);
            CHECK(ctx, ngsq_get_edits_positions(ctx, r, refs.data(), alts.data(), L1));
            for (size_t i = 0; i < L1; i++) {
                const uint64_t total = (uint64_t)refs[i] + alts[i];
                if (!total) continue;
                const float vaf = (float)alts[i] / (float)total;
                char num[64];
                const auto res = std::to_chars(num, num + sizeof num, vaf, std::chars_format::fixed);
                fprintf(vaf_file, "%s\t%zu\t%.*s\n", names[r].c_str(), i, (int)(res.ptr - num), num);
            }
        }
    }
    if (vaf_file) fclose(vaf_file);

    logf(2, "Aggregating results.");
    std::vector<const char *> name_ptrs(n_refs ? n_refs : 1, "");
    for (uint32_t r = 0; r < n_refs; r++) name_ptrs[r] = names[r].c_str();
    const int64_t need = ngsq_results_json(ctx, name_ptrs.data(), nullptr, 0);
    if (need < 0) bail("could not serialize results");
    std::vector<char> buf((size_t)need + 1);
    ngsq_results_json(ctx, name_ptrs.data(), buf.data(), buf.size());
    logf(2, "Writing output.");
    const std::string out_path = a.out_dir + "/" + a.prefix + ".results.json"; // results.rs:50-60
    FILE *of = fopen(out_path.c_str(), "wb");
    if (!of || fwrite(buf.data(), 1, (size_t)need, of) != (size_t)need) bail("could not write " + out_path);
    fclose(of);
    milestone("results written");
    if (comm) ngsq_comm_barrier(comm); // nobody leaves while a rank may still be reading its messages
    if (quick_exit) {
        // everything is on disk: leave the GiB of device and pinned memory to the kernel's process teardown instead of
        // unmapping them block by block and running the HIP runtime's exit handlers (measurement: DESIGN.md section 7)
        if (comm) ngsq_comm_destroy(comm); // rank 0 unlinks the shared-memory segment
        fflush(nullptr);
        report_done();
        _exit(0);
    }
    ngsq_destroy(ctx);
    ngsq_bam_close(bam);
    ngsq_fasta_close(fasta);
    milestone("context and reader released");
    if (comm) ngsq_comm_destroy(comm);
    return 0;
}
