// host_util.cpp -- the master-mode / loader pieces of the path that run on the host in Ray as well
// (they touch tens of histogram bins or one read at a time; there is nothing to accelerate).
#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/raygraph.h"

extern "C" {

// CoverageDistribution::CoverageDistribution (CoverageGatherer/CoverageDistribution.cpp:33-112).
// File format: two '#' header lines then "<coverage> <count>\n" ascending (:39-44).  Peak: every
// position votes for the largest y inside a 10-wide window to its right (:55-85); the winner has
// the most votes or the largest y (:88-93); minimum = lowest y at or left of the peak, leftmost on
// ties (:95-103); repeat = min(2*peak, peak + (peak - minimum)) (:106-111).
int rg_coverage_distribution(const uint32_t *coverage, const uint64_t *count, uint64_t n, const char *path,
                             int32_t *minimum, int32_t *peak, int32_t *repeat) {
	if ((n && (!coverage || !count)) || !minimum || !peak || !repeat) return RG_ERR_ARG;
	if (path) {
		FILE *f = fopen(path, "w");
		if (!f) return RG_ERR_ARG;
		fputs("# KmerCoverage\tFrequency\n", f);
		fputs("# Any frequency is a even number because of odd k-mer length\n", f);
		for (uint64_t i = 0; i < n; ++i) fprintf(f, "%u %llu\n", coverage[i], (unsigned long long)count[i]);
		fclose(f);
	}
	if (n == 0) return RG_ERR_ARG;  // "Assembler panic: no k-mers found in reads" (MachineHelper.cpp:261-268)

	const int size = (int)n;
	const int windowSize = 10;
	const uint32_t minimumX = 1, maximumX = 65535 - 1, safeThreshold = 256;
	const uint64_t minimumY = 2 * 4096, minimumY2 = 55000;

	std::map<int, int> votes;
	for (int i = 0; i < size; ++i) {
		int best = i;
		for (int j = 0; j < windowSize && i + j < size; ++j)
			if (count[i + j] > count[best]) best = i + j;
		if (coverage[best] > maximumX || coverage[best] < minimumX) continue;
		if (coverage[best] >= safeThreshold && count[best] < minimumY2) continue;
		if (count[best] > minimumY) votes[best]++;
	}
	// With no vote the reference dereferences begin() of an empty map (undefined); position 0 is what
	// that read yields with libstdc++ and is what we define here.
	int largest = votes.empty() ? 0 : votes.begin()->first;
	for (std::map<int, int>::iterator it = votes.begin(); it != votes.end(); ++it)
		if (it->second > votes[largest] || count[it->first] > count[largest]) largest = it->first;

	int lowest = largest;
	for (int i = largest; i >= 0; --i)
		if (count[i] <= count[lowest]) lowest = i;

	const int minCov = (int)coverage[lowest], peakCov = (int)coverage[largest];
	int repeatCov = 2 * peakCov;
	const int candidate = peakCov + (peakCov - minCov);
	if (candidate < repeatCov) repeatCov = candidate;
	*minimum = minCov;
	*peak = peakCov;
	*repeat = repeatCov;
	return RG_OK;
}

// Read::constructor with Read::trim (SequencesLoader/Read.cpp:31-159): lower-case acgt are
// upper-cased, leading/trailing non-ACGT symbols are trimmed, inner ones become A; base i goes to
// bits 2(i%4).. of byte i/4 with A=0 C=1 G=2 T=3 (Mock/constants.h:79-82).
int rg_pack_read(const char *sequence, int trim, uint8_t *packed) {
	if (!sequence || !packed) return RG_ERR_ARG;
	std::string s(sequence);
	auto is_acgt = [](char c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T'; };
	if (trim && s.size() < 65535) {
		for (char &c : s) {
			if (c == 'a') c = 'A';
			else if (c == 't') c = 'T';
			else if (c == 'c') c = 'C';
			else if (c == 'g') c = 'G';
		}
		size_t first = 0;
		while (first < s.size() && !is_acgt(s[first])) first++;
		size_t last = first;  // one past the last ACGT symbol
		for (size_t i = first; i < s.size(); ++i)
			if (is_acgt(s[i])) last = i + 1;
		s = s.substr(first, last - first);
	}
	const size_t bytes = (s.size() + 3) / 4;
	memset(packed, 0, bytes);
	for (size_t i = 0; i < s.size(); ++i) {
		uint8_t code = 0;
		switch (s[i]) {
		case 'C': code = 1; break;
		case 'G': code = 2; break;
		case 'T': code = 3; break;
		default: code = 0;  // A and anything else (Read.cpp:117-124)
		}
		packed[i / 4] |= (uint8_t)(code << ((i % 4) * 2));
	}
	return (int)s.size();
}


// ---- on-disk formats of the path ---------------------------------------------------

}  // extern "C"
namespace {

struct K2 {
	uint64_t w[2];
};

// Kmer::complementVertex (KmerAcademyBuilder/Kmer.cpp:398-420), word-parallel
K2 revcomp_host(K2 a, int k, int W) {
	auto rev2 = [](uint64_t x) {  // reverses the order of the 32 two-bit groups
		x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
		x = ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
		return __builtin_bswap64(x);
	};
	K2 r{{0, 0}};
	if (W == 1) {
		r.w[0] = (~rev2(a.w[0])) >> (64 - 2 * k);
		return r;
	}
	// 128-bit string reversed and complemented: low word from word 1, high word from word 0
	const uint64_t lo = ~rev2(a.w[1]), hi = ~rev2(a.w[0]);
	const int sh = 128 - 2 * k;
	if (sh >= 64) {
		r.w[0] = hi >> (sh - 64);
	} else {
		r.w[0] = (lo >> sh) | (hi << (64 - sh));
		r.w[1] = hi >> sh;
	}
	return r;
}

// Vertex::convertBitmap (Vertex.cpp:341-361)
uint8_t convert_bitmap_host(uint8_t m) {
	uint32_t s = ((m << 4) | (m >> 4)) & 0xFFu;
	s = ((s & 0xAAu) >> 1) | ((s & 0x55u) << 1);
	s = ((s & 0xCCu) >> 2) | ((s & 0x33u) << 2);
	return (uint8_t)s;
}

// Kmer::getIngoingEdges / getOutgoingEdges (Kmer.cpp:176-282): neighbour named by base c
K2 parent_of(K2 a, int c, int k, int W) {
	K2 r;
	r.w[0] = (a.w[0] << 2) | (uint64_t)c;
	r.w[1] = (W == 2) ? ((a.w[1] << 2) | (a.w[0] >> 62)) : 0;
	if (2 * k < 64) r.w[0] &= (1ull << (2 * k)) - 1;
	else if (2 * k > 64) r.w[1] &= (1ull << (2 * k - 64)) - 1;
	if (2 * k <= 64) r.w[1] = 0;
	return r;
}
K2 child_of(K2 a, int c, int k, int W) {
	K2 r;
	r.w[0] = a.w[0] >> 2;
	r.w[1] = 0;
	if (W == 2) {
		r.w[0] |= a.w[1] << 62;
		r.w[1] = a.w[1] >> 2;
	}
	const int pos = 2 * (k - 1);
	if (pos < 64) r.w[0] |= (uint64_t)c << pos;
	else r.w[1] |= (uint64_t)c << (pos - 64);
	return r;
}

}  // namespace
extern "C" {

// GenomeGraph checkpoint (MessageProcessor.cpp:658-678 writes it, :596-655 reads it): LargeCount n =
// GridTable::size() (2 per stored pair), then per iterator step -- the stored key, then its reverse
// complement (GridTableIterator.cpp:41-54) -- Vertex::write (Vertex.cpp:257-273): key words, int coverage,
// int #parents + their words, int #children + their words (bases A,C,G,T in that order).
int rg_write_genome_graph(const void *records, uint64_t n_pairs, int k, int kmer_words, const char *path) {
	if ((n_pairs && !records) || !path || k < 1 || k > 64 || (kmer_words != 1 && kmer_words != 2)) return RG_ERR_ARG;
	if ((kmer_words == 1 && k > 32)) return RG_ERR_ARG;
	FILE *f = fopen(path, "wb");
	if (!f) return RG_ERR_ARG;
	const int W = kmer_words;
	const size_t rec = (size_t)W * 8 + 5;
	const uint64_t theSize = 2 * n_pairs;
	fwrite(&theSize, sizeof theSize, 1, f);
	const uint8_t *p = (const uint8_t *)records;
	for (uint64_t i = 0; i < n_pairs; ++i, p += rec) {
		K2 key{{0, 0}};
		memcpy(key.w, p, (size_t)W * 8);
		uint32_t cov;
		memcpy(&cov, p + W * 8, 4);
		const uint8_t stored = p[W * 8 + 4];
		K2 keys[2] = {key, revcomp_host(key, k, W)};
		for (int s = 0; s < 2; ++s) {
			const int coverage = (int)cov;
			// Vertex::getEdges (Vertex.cpp:169-173): the stored byte is in the lower key's orientation
			const uint8_t e = (s == 0) ? stored : convert_bitmap_host(stored);
			fwrite(keys[s].w, 8, (size_t)W, f);
			fwrite(&coverage, sizeof(int), 1, f);
			int np = 0, nc = 0;
			for (int c = 0; c < 4; ++c) {
				np += (e >> c) & 1;
				nc += (e >> (4 + c)) & 1;
			}
			fwrite(&np, sizeof(int), 1, f);
			for (int c = 0; c < 4; ++c)
				if ((e >> c) & 1) {
					K2 nb = parent_of(keys[s], c, k, W);
					fwrite(nb.w, 8, (size_t)W, f);
				}
			fwrite(&nc, sizeof(int), 1, f);
			for (int c = 0; c < 4; ++c)
				if ((e >> (4 + c)) & 1) {
					K2 nb = child_of(keys[s], c, k, W);
					fwrite(nb.w, 8, (size_t)W, f);
				}
		}
	}
	const bool ok = !ferror(f);
	fclose(f);
	return ok ? RG_OK : RG_ERR_ARG;
}

// CoverageDistribution checkpoint (MachineHelper.cpp:243-259 writes it, :226-241 reads it)
int rg_write_coverage_checkpoint(const uint32_t *coverage, const uint64_t *count, uint64_t n, const char *path) {
	if ((n && (!coverage || !count)) || !path) return RG_ERR_ARG;
	FILE *f = fopen(path, "wb");
	if (!f) return RG_ERR_ARG;
	const int theSize = (int)n;
	fwrite(&theSize, sizeof(int), 1, f);
	for (uint64_t i = 0; i < n; ++i) {
		fwrite(&coverage[i], sizeof(uint32_t), 1, f);
		fwrite(&count[i], sizeof(uint64_t), 1, f);
	}
	fclose(f);
	return RG_OK;
}

// CoverageDistributionAnalysis.txt (MachineHelper.cpp:283-327): written by the master right after CoverageDistribution.txt.
// Lowest coverage = the smallest key below 9999 (the loop's start value, :285,295-298); the percentage is a double printed
// by a C++ stream = %g.
int rg_write_coverage_analysis(const uint32_t *coverage, const uint64_t *count, uint64_t n, int k, int32_t minimum, int32_t peak,
                               int32_t repeat, const char *distribution_file, const char *path) {
	if ((n && (!coverage || !count)) || !distribution_file || !path) return RG_ERR_ARG;
	uint64_t vertices = 0, lowest_count = 0, at_least_minimum = 0;
	uint32_t lowest = 9999;
	for (uint64_t i = 0; i < n; ++i) {
		if (coverage[i] < lowest) {
			lowest = coverage[i];
			lowest_count = count[i];
		}
		if ((int64_t)coverage[i] >= (int64_t)minimum) at_least_minimum += count[i];
		vertices += count[i];
	}
	FILE *f = fopen(path, "w");
	if (!f) return RG_ERR_ARG;
	fprintf(f, "k-mer length:\t%d\n", k);
	fprintf(f, "Number of k-mers in the distributed de Bruijn graph: %llu\n", (unsigned long long)vertices);
	fprintf(f, "Lowest coverage observed:\t%u\n", lowest);
	fprintf(f, "MinimumCoverage:\t%d\nPeakCoverage:\t%d\nRepeatCoverage:\t%d\n", minimum, peak, repeat);
	fprintf(f, "Number of k-mers with at least MinimumCoverage:\t%llu k-mers\n", (unsigned long long)at_least_minimum);
	fprintf(f, "Percentage of vertices with coverage %u:\t%g %%\n", lowest, (0.0 + (double)lowest_count) / (double)vertices * 100.00);
	fprintf(f, "DistributionFile: %s\n", distribution_file);
	fclose(f);
	return RG_OK;
}

// The reader of Rank<r>.Checkpoint.GenomeGraph.ray, shaped like the reference's (MessageProcessor.cpp:596-655): every entry is
// a k-mer of either strand with its coverage, its parents and its children as whole k-mers.  The reference inserts the
// k-mer's lower key, takes the coverage only from the entry that creates the vertex and only if that entry IS the lower key
// (Vertex::setCoverage, Vertex.cpp:80-92: the writer emits the lower key first, GridTableIterator.cpp:41-54), and adds one
// edge bit per parent / child through Vertex::addIngoingEdge / addOutgoingEdge (first base of the parent, last base of the
// child, in the entry's orientation, converted to the lower key's).  Output: Vertex::dump records (what rg_import_vertices
// and Vertex::load take), one per vertex pair, in the order the pairs first appear.  The CONFIG_ASSERT checks of :633-641,650
// are errors here: a parent / child that is not a neighbour of its k-mer, or a count that is not 2 per pair -> RG_ERR_ARG.
// records == NULL: *n_pairs = the header's count / 2, nothing else is read.
int rg_read_genome_graph(const char *path, int k, int kmer_words, void *records, uint64_t cap_pairs, uint64_t *n_pairs) {
	if (!path || !n_pairs || k < 1 || k > 64 || (kmer_words != 1 && kmer_words != 2) || (kmer_words == 1 && k > 32)) return RG_ERR_ARG;
	FILE *f = fopen(path, "rb");
	if (!f) return RG_ERR_ARG;
	const int W = kmer_words;
	const size_t rec = (size_t)W * 8 + 5;
	uint64_t n = 0;
	if (fread(&n, sizeof n, 1, f) != 1 || (n & 1)) {
		fclose(f);
		return RG_ERR_ARG;
	}
	*n_pairs = n / 2;
	if (!records) {
		fclose(f);
		return RG_OK;
	}
	if (cap_pairs < n / 2) {
		fclose(f);
		return RG_ERR_CAPACITY;
	}
	std::vector<char> iobuf(4u << 20);
	setvbuf(f, iobuf.data(), _IOFBF, iobuf.size());
	auto less = [&](const K2 &a, const K2 &b) { return a.w[0] != b.w[0] ? a.w[0] < b.w[0] : a.w[1] < b.w[1]; };  // Kmer::operator< (Kmer.cpp:38-47)
	auto same = [&](const K2 &a, const K2 &b) { return a.w[0] == b.w[0] && a.w[1] == b.w[1]; };
	auto read_kmer = [&](K2 *out) {
		out->w[0] = out->w[1] = 0;
		return fread(out->w, 8, (size_t)W, f) == (size_t)W;
	};
	uint8_t *out = (uint8_t *)records;
	uint64_t made = 0;
	std::map<std::pair<uint64_t, uint64_t>, uint64_t> seen;  // lower key -> record index, only for pairs that do not arrive back to back
	K2 last_lower{{~0ull, ~0ull}};
	uint64_t last_index = 0;
	bool have_last = false, ok = true;
	for (uint64_t i = 0; i < n && ok; ++i) {
		K2 kmer;
		int coverage = 0, parents = 0, children = 0;
		if (!read_kmer(&kmer) || fread(&coverage, sizeof(int), 1, f) != 1) {
			ok = false;
			break;
		}
		const K2 rc = revcomp_host(kmer, k, W);
		const bool is_lower = !less(rc, kmer);
		const K2 lower = is_lower ? kmer : rc;
		uint64_t idx;
		bool inserted = false;
		if (have_last && same(lower, last_lower)) idx = last_index;
		else {
			auto it = seen.find({lower.w[0], lower.w[1]});
			if (it != seen.end()) idx = it->second;
			else {
				if (made >= cap_pairs) {  // more than count / 2 distinct pairs: the header lies (the reader's size assertion, :650)
					ok = false;
					break;
				}
				if (have_last) seen[{last_lower.w[0], last_lower.w[1]}] = last_index;
				idx = made++;
				inserted = true;
				uint8_t *p = out + idx * rec;
				memcpy(p, lower.w, (size_t)W * 8);
				memset(p + W * 8, 0, 5);
			}
		}
		last_lower = lower;
		last_index = idx;
		have_last = true;
		uint8_t *p = out + idx * rec;
		if (inserted && is_lower) {  // tmp->constructor(); tmp->setCoverage(&kmer, coverage)
			const uint32_t c = (uint32_t)coverage;
			memcpy(p + W * 8, &c, 4);
		}
		uint8_t e = 0;  // in the entry's orientation
		if (fread(&parents, sizeof(int), 1, f) != 1 || parents < 0 || parents > 4) ok = false;
		for (int j = 0; j < parents && ok; ++j) {
			K2 nb;
			if (!read_kmer(&nb)) ok = false;
			else {
				const int c = (int)(nb.w[0] & 3);  // Kmer::getFirstSegmentFirstCode (Kmer.cpp:155-163)
				if (!same(parent_of(kmer, c, k, W), nb) || ((e >> c) & 1)) ok = false;  // :633-639
				e |= (uint8_t)(1u << c);
			}
		}
		if (ok && (fread(&children, sizeof(int), 1, f) != 1 || children < 0 || children > 4)) ok = false;
		for (int j = 0; j < children && ok; ++j) {
			K2 nb;
			if (!read_kmer(&nb)) ok = false;
			else {
				const int pos = 2 * (k - 1);
				const int c = (int)((nb.w[pos / 64] >> (pos % 64)) & 3);  // Kmer::getSecondSegmentLastCode (Kmer.cpp:145-153)
				if (!same(child_of(kmer, c, k, W), nb) || ((e >> (4 + c)) & 1)) ok = false;  // :640
				e |= (uint8_t)(1u << (4 + c));
			}
		}
		if (ok) p[W * 8 + 4] |= is_lower ? e : convert_bitmap_host(e);
	}
	fclose(f);
	if (!ok || made * 2 != n) return RG_ERR_ARG;
	return RG_OK;
}

// EdgePurger::writeGraphPartition (EdgePurger.cpp:172-217): sizes[r] = GridTable::size() of rank r
int rg_write_graph_partition(const uint64_t *sizes, int nranks, const char *path) {
	if (!sizes || nranks < 1 || !path) return RG_ERR_ARG;
	FILE *f = fopen(path, "w");
	if (!f) return RG_ERR_ARG;
	uint64_t total = 0;
	for (int i = 0; i < nranks; ++i) total += sizes[i];
	const uint64_t best = total / (uint64_t)nranks;
	fprintf(f, "#Rank\tNumberOfKmers\tIdealNumberOfKmers\tDifference\tRelativeDifference\n");
	fprintf(f, "#TotalKmers: %llu\n", (unsigned long long)total);
	fprintf(f, "#Ranks: %d\n", nranks);
	fprintf(f, "#IdealNumberOfKmers: %llu\n", (unsigned long long)best);
	for (int i = 0; i < nranks; ++i) {
		const uint64_t actual = sizes[i];
		int difference = (actual < best) ? -(int)(best - actual) : (int)(actual - best);
		double relativeDifference = difference * 100;
		if (best != 0) relativeDifference /= (double)best;
		fprintf(f, "%d\t%llu\t%llu\t%d\t%g%%\n", i, (unsigned long long)actual, (unsigned long long)best, difference, relativeDifference);
	}
	fclose(f);
	return RG_OK;
}

// kmers.txt: CoverageGatherer::writeKmers (CoverageGatherer/CoverageGatherer.cpp:47-122) with the header of
// :202-228 (written by the master rank only, :70-71).  One line per strand of every vertex:
// SEQ;coverage;first base of each parent;last base of each child (A,C,G,T order, Kmer.cpp:200-211,269-280).
int rg_write_kmers_txt(const void *records, uint64_t n_pairs, int k, int kmer_words, const char *path, int write_header,
                       const char *ray_version, int append) {
	if ((n_pairs && !records) || !path || k < 1 || k > 64 || (kmer_words != 1 && kmer_words != 2)) return RG_ERR_ARG;
	FILE *f = fopen(path, append ? "a" : "w");
	if (!f) return RG_ERR_ARG;
	if (write_header) {
		fprintf(f, "# Generated by Ray %s\n", ray_version ? ray_version : "");
		fprintf(f, "# The length of k-mers is %i\n", k);
		fputs("# This file contains the k-mer graph (subgraph of the de Bruijn graph for alphabet={A,C,G,T})\n", f);
		fputs("# The genome sequence you are looking for is a path in this maze\n", f);
		fputs("# You can kickstart your assembly algorithm development by loading this file\n", f);
		fputs("# Note that vertices with a coverage of 1 are not considered.\n", f);
		fputs("# Format:\n", f);
		fputs("# k-mer nucleotide sequence; coverage value; first nucleotide of parents; last nucleotide of children\n", f);
		fputs("# Example in nucleotide space:\n", f);
		fputs("# ATCG;10;T G;C A\n", f);
		fputs("#  ATCG has a coverage value of 10\n", f);
		fputs("#  Ingoing arcs: TATC -> ATCG and GATC -> ATCG\n", f);
		fputs("#  Outgoing arcs: ATCG -> TCGC and ATCG -> TCGA\n", f);
	}
	const int W = kmer_words;
	const size_t rec = (size_t)W * 8 + 5;
	const uint8_t *p = (const uint8_t *)records;
	static const char CODES[] = "ACGT";
	char line[256];
	for (uint64_t i = 0; i < n_pairs; ++i, p += rec) {
		K2 key{{0, 0}};
		memcpy(key.w, p, (size_t)W * 8);
		uint32_t cov;
		memcpy(&cov, p + W * 8, 4);
		const uint8_t stored = p[W * 8 + 4];
		K2 keys[2] = {key, revcomp_host(key, k, W)};
		for (int s = 0; s < 2; ++s) {
			const uint8_t e = (s == 0) ? stored : convert_bitmap_host(stored);
			int len = 0;
			for (int j = 0; j < k; ++j) line[len++] = CODES[(keys[s].w[j / 32] >> (2 * (j % 32))) & 3];  // Kmer::idToWord (Kmer.cpp:312-336)
			len += snprintf(line + len, sizeof line - (size_t)len, ";%u;", cov);
			bool firstOne = true;
			for (int c = 0; c < 4; ++c)
				if ((e >> c) & 1) {
					if (!firstOne) line[len++] = ' ';
					line[len++] = CODES[c];
					firstOne = false;
				}
			line[len++] = ';';
			firstOne = true;
			for (int c = 0; c < 4; ++c)
				if ((e >> (4 + c)) & 1) {
					if (!firstOne) line[len++] = ' ';
					line[len++] = CODES[c];
					firstOne = false;
				}
			line[len++] = '\n';
			fwrite(line, 1, (size_t)len, f);
		}
	}
	fclose(f);
	return RG_OK;
}

// ---- read ingest -------------------------------------------------------------------

}  // extern "C"
namespace {

bool ends_with(const std::string &s, const char *suffix) {
	const size_t n = strlen(suffix);
	return s.size() >= n && s.compare(s.size() - n, n, suffix) == 0;
}
// LoaderFactory / FastaLoader::checkFileType, FastqLoader::checkFileType: by file name
int file_kind(const char *path) {
	const std::string s(path);
	if (ends_with(s, ".fastq") || ends_with(s, ".fq")) return 1;
	if (ends_with(s, ".fasta") || ends_with(s, ".fa") || ends_with(s, ".fna") || ends_with(s, ".fas")) return 0;
	return -1;
}

// walks the records of a file; emit(sequence text, length) per record, in file order
template <class F>
int walk_sequences(const char *path, int kind, F emit) {
	FILE *f = fopen(path, "r");
	if (!f) return RG_ERR_ARG;
	std::vector<char> line(65536 + 8);
	if (kind == 1) {  // FastqLoader::loadWithPeriod (FastqLoader.cpp:84-119): line 1 of every 4
		int rotating = 0;
		while (fgets(line.data(), 65535, f)) {
			if (rotating == 1) emit(line.data(), strlen(line.data()));
			if (++rotating == 4) rotating = 0;
		}
	} else {  // FastaLoader::load (FastaLoader.cpp:26-64): tokens; '>' opens a record, rest of that line skipped
		std::string seq;
		bool have_id = false;
		char token[65536 + 8];
		while (fscanf(f, "%65535s", token) == 1) {
			if (token[0] == '>') {
				int c;
				while ((c = fgetc(f)) != EOF && c != '\n') {
				}
				if (have_id) emit(seq.c_str(), seq.size());
				have_id = true;
				seq.clear();
			} else {
				seq += token;
			}
		}
		emit(seq.c_str(), seq.size());  // the last record is pushed unconditionally (:56-61)
	}
	fclose(f);
	return RG_OK;
}

}  // namespace
extern "C" {

int rg_count_sequences(const char *path, uint64_t *n) {
	if (!path || !n) return RG_ERR_ARG;
	const int kind = file_kind(path);
	if (kind < 0) return RG_ERR_ARG;
	uint64_t c = 0;
	int rc = walk_sequences(path, kind, [&](const char *, size_t) { c++; });
	*n = c;
	return rc;
}

// sequences [first, last) of the file go to the device in slabs and are trimmed and packed there
int rg_load_sequences(rg_handle *h, const char *path, uint64_t first, uint64_t last, uint64_t *loaded) {
	if (!h || !path) return RG_ERR_ARG;
	const int kind = file_kind(path);
	if (kind < 0) return RG_ERR_ARG;
	const size_t SLAB = 256u << 20;
	std::vector<char> text;
	std::vector<uint64_t> starts;
	std::vector<uint32_t> lens;
	text.reserve(std::min<size_t>(SLAB + 65536, 64u << 20));
	uint64_t idx = 0, done = 0;
	int status = RG_OK;
	auto flush = [&]() {
		if (!starts.empty() && status == RG_OK) status = rg_add_reads_text(h, text.data(), text.size(), starts.data(), lens.data(), starts.size());
		done += starts.size();
		text.clear();
		starts.clear();
		lens.clear();
	};
	int rc = walk_sequences(path, kind, [&](const char *s, size_t len) {
		const uint64_t i = idx++;
		if (i < first || i >= last || status != RG_OK) return;
		starts.push_back(text.size());
		lens.push_back((uint32_t)std::min<size_t>(len, 0xFFFFFFFFu));
		text.insert(text.end(), s, s + len);
		if (text.size() >= SLAB) flush();
	});
	if (rc != RG_OK) return rc;
	flush();
	if (loaded) *loaded = done;
	return status;
}

}  // extern "C"
