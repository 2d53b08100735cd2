// raygraph_build -- stand-alone driver over the C ABI: the "no source change" boundary of SURVEY.md 8(b).
// Reads FASTA/FASTQ files the way Ray's command line names them, builds the k-mer graph on one MI355X and
// writes what Ray's graph-construction steps write: <prefix>CoverageDistribution.txt, <prefix>kmers.txt
// (-write-kmers), <prefix>degreeDistribution.txt, <prefix>GraphPartition.txt and, with -write-checkpoints DIR,
// Rank0.Checkpoint.GenomeGraph.ray + Rank0.Checkpoint.CoverageDistribution.ray, from which a stock Ray started
// with -read-checkpoints DIR resumes after graph construction (MessageProcessor.cpp:596-655,
// MachineHelper.cpp:226-241).
//
//   raygraph_build -k 31 -p left.fastq right.fastq -s single.fasta -o out/ [-write-kmers]
//                  [-write-checkpoints DIR] [-bloom-filter-bits N] [-device D]
#include <raygraph.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static double now() {
	return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

#define CHECK(call)                                                                              \
	do {                                                                                         \
		int rc_ = (call);                                                                        \
		if (rc_ != RG_OK) {                                                                      \
			fprintf(stderr, "Rank 0: %s failed (%d): %s\n", #call, rc_, rg_last_error(h));       \
			if (h) rg_destroy(h);                                                                \
			return 1;                                                                            \
		}                                                                                        \
	} while (0)

int main(int argc, char **argv) {
	int k = 21, device = 0;  // Ray's default k-mer length (Mock/Parameters.cpp:79)
	uint64_t bloom = RG_BLOOM_DEFAULT;
	std::string prefix = "RayOutput/", ckpt;
	bool write_kmers = false;
	std::vector<std::string> files;  // in command-line order: read ids run through them in this order (SequencesLoader.cpp:302-359)
	for (int i = 1; i < argc; ++i) {
		const std::string a = argv[i];
		auto need = [&](int n) {
			if (i + n >= argc) {
				fprintf(stderr, "%s needs %d value(s)\n", a.c_str(), n);
				exit(2);
			}
		};
		if (a == "-k") { need(1); k = atoi(argv[++i]); }
		else if (a == "-p") { need(2); files.push_back(argv[++i]); files.push_back(argv[++i]); }
		else if (a == "-s") { need(1); files.push_back(argv[++i]); }
		else if (a == "-o") { need(1); prefix = argv[++i]; }
		else if (a == "-write-kmers") write_kmers = true;
		else if (a == "-write-checkpoints") { need(1); ckpt = argv[++i]; }
		else if (a == "-bloom-filter-bits") { need(1); bloom = strtoull(argv[++i], nullptr, 10); }
		else if (a == "-device") { need(1); device = atoi(argv[++i]); }
		else {
			fprintf(stderr, "unknown option %s\n", a.c_str());
			return 2;
		}
	}
	if (files.empty() || k < 1 || k > 64) {
		fprintf(stderr, "usage: raygraph_build -k K (-p left right | -s file)... -o prefix [-write-kmers] [-write-checkpoints DIR] [-bloom-filter-bits N]\n");
		return 2;
	}
	if (k % 2 == 0) {  // Mock/Parameters.cpp:1037-1039
		k--;
		printf("Warning: k-mer length must be odd, using %d\n", k);
	}

	rg_handle *h = nullptr;
	rg_config cfg;
	memset(&cfg, 0, sizeof cfg);
	cfg.k = k;
	cfg.kmer_words = 0;
	cfg.rank = 0;
	cfg.size = 1;
	cfg.device = device;
	cfg.bloom_bits = bloom;
	CHECK(rg_create(&cfg, &h));

	double t0 = now();
	uint64_t total = 0;
	for (const std::string &f : files) {
		uint64_t n = 0;
		CHECK(rg_load_sequences(h, f.c_str(), 0, UINT64_MAX, &n));
		printf("Rank 0 has %llu sequence reads from %s\n", (unsigned long long)n, f.c_str());
		total += n;
	}
	double t1 = now();
	CHECK(rg_count_kmers(h));  // RAY_SLAVE_MODE_ADD_VERTICES
	double t2 = now();
	uint64_t nh = 0;
	CHECK(rg_histogram(h, nullptr, nullptr, 0, &nh));  // RAY_SLAVE_MODE_SEND_DISTRIBUTION
	std::vector<uint32_t> cov(nh ? nh : 1);
	std::vector<uint64_t> cnt(nh ? nh : 1);
	CHECK(rg_histogram(h, cov.data(), cnt.data(), nh, &nh));
	int32_t minimum = 0, peak = 0, repeat = 0;
	const std::string dist = prefix + "CoverageDistribution.txt";
	if (rg_coverage_distribution(cov.data(), cnt.data(), nh, dist.c_str(), &minimum, &peak, &repeat) != RG_OK) {
		fprintf(stderr, "Rank 0: Assembler panic: no k-mers found in reads.\n");  // MachineHelper.cpp:261-268
		rg_destroy(h);
		return 1;
	}
	CHECK(rg_add_edges(h));    // RAY_SLAVE_MODE_ADD_EDGES
	CHECK(rg_purge_edges(h));  // RAY_SLAVE_MODE_PURGE_NULL_EDGES
	double t3 = now();

	uint64_t pairs = 0, size2 = 0;
	CHECK(rg_export(h, nullptr, 0, &pairs));
	CHECK(rg_size(h, &size2));
	rg_stats st;
	CHECK(rg_get_stats(h, &st));
	const int W = (k <= 31) ? 1 : 2;  // what rg_create picks for kmer_words = 0
	const size_t rec = (size_t)W * 8 + 5;
	std::vector<char> records((size_t)std::max<uint64_t>(1, pairs) * rec);
	CHECK(rg_export(h, records.data(), pairs, &pairs));
	uint64_t degrees[25];
	CHECK(rg_degree_distribution(h, degrees));

	if (write_kmers && rg_write_kmers_txt(records.data(), pairs, k, W, (prefix + "kmers.txt").c_str(), 1, "raygraph", 0) != RG_OK) {
		fprintf(stderr, "cannot write %skmers.txt\n", prefix.c_str());
		return 1;
	}
	{  // degreeDistribution.txt (MachineHelper.cpp: RAY_MPI_TAG_WRITE_KMERS_REPLY accumulation)
		FILE *f = fopen((prefix + "degreeDistribution.txt").c_str(), "w");
		if (f) {
			fputs("# Most of the vertices should have an ingoing degree of 1 and an outgoing degree of 1.\n", f);
			fputs("# These are the easy vertices.\n", f);
			fputs("# Then, the most abundant are those with an ingoing degree of 1 and an outgoing degree of 2.\n", f);
			fputs("# Note that vertices with a coverage of 1 are not considered.\n", f);
			fputs("# The option -write-kmers will actually write all the graph to a file if you need more precise data.\n", f);
			fputs("# IngoingDegree\tOutgoingDegree\tNumberOfVertices\n", f);
			for (int i = 0; i <= 4; ++i)
				for (int j = 0; j <= 4; ++j) fprintf(f, "%d\t%d\t%llu\n", i, j, (unsigned long long)degrees[i * 5 + j]);
			fclose(f);
		}
	}
	rg_write_graph_partition(&size2, 1, (prefix + "GraphPartition.txt").c_str());
	if (!ckpt.empty()) {
		if (rg_write_genome_graph(records.data(), pairs, k, W, (ckpt + "/Rank0.Checkpoint.GenomeGraph.ray").c_str()) != RG_OK ||
		    rg_write_coverage_checkpoint(cov.data(), cnt.data(), nh, (ckpt + "/Rank0.Checkpoint.CoverageDistribution.ray").c_str()) != RG_OK) {
			fprintf(stderr, "cannot write checkpoints into %s\n", ckpt.c_str());
			return 1;
		}
	}
	printf("Rank 0: %llu reads, %llu k-mer occurrences, %llu vertices (both strands)\n", (unsigned long long)total,
	       (unsigned long long)st.occurrences, (unsigned long long)size2);
	printf("Rank 0: minimum coverage %d, peak coverage %d, repeat coverage %d\n", minimum, peak, repeat);
	printf("Rank 0: sequence loading %.3f s, k-mer counting %.3f s, distribution + edges + purge %.3f s\n", t1 - t0, t2 - t1, t3 - t2);
	rg_destroy(h);
	return 0;
}
