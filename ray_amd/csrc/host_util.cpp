// host_util.cpp -- the master-mode / loader pieces of the path that run on the host in Ray as well
// (they touch tens of histogram bins or one read at a time; there is nothing to accelerate).
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

}  // extern "C"
