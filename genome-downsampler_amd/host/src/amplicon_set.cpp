#include "bam-api/amplicon_set.hpp"

#include <fstream>
#include <sstream>
#include <stdexcept>

namespace bam_api {

bool read_primer_bed(const std::filesystem::path& path, PrimerMap& out) {
    std::ifstream file(path);
    if (!file.is_open()) return false;
    std::string line;
    while (std::getline(file, line)) {
        std::istringstream fields(line);
        std::string chrom, start, end, name;
        std::getline(fields, chrom, '\t');
        std::getline(fields, start, '\t');
        std::getline(fields, end, '\t');
        std::getline(fields, name, '\t');
        Index s = 0, e = 0;
        try {
            s = std::stoull(start);
            e = std::stoull(end);
        } catch (const std::invalid_argument&) {
            continue;
        } catch (const std::out_of_range&) {
            continue;
        }
        if (chrom.empty() || start.empty() || end.empty() || name.empty()) continue;
        out.emplace(name, std::make_pair(s, e));  // emplace: an existing name is kept
    }
    return true;
}

bool read_primer_pairs_tsv(const std::filesystem::path& path,
                           std::vector<std::pair<std::string, std::string>>& out) {
    std::ifstream file(path);
    if (!file.is_open()) return false;
    std::string line;
    while (std::getline(file, line)) {
        std::istringstream fields(line);
        std::string left, right;
        std::getline(fields, left, '\t');
        std::getline(fields, right, '\t');
        if (!left.empty() && !right.empty()) out.emplace_back(left, right);
    }
    return true;
}

AmpliconSet build_amplicon_set(PrimerMap primers,
                               const std::vector<std::pair<std::string, std::string>>* pairs) {
    AmpliconSet set;
    auto add = [&set](std::pair<Index, Index>& left, std::pair<Index, Index>& right) {
        if (left.first > right.first) std::swap(left, right);
        set.amplicons.emplace_back(left.first, right.second);
    };
    if (pairs != nullptr) {
        for (const auto& names : *pairs) add(primers[names.first], primers[names.second]);
    } else {
        // an odd trailing primer has no partner (the reference walks past the end there)
        for (auto it = primers.begin(); it != primers.end();) {
            auto& left = it->second;
            if (++it == primers.end()) break;
            add(left, it->second);
            ++it;
        }
    }
    return set;
}

bool amplicon_set_from_files(const std::filesystem::path& bed, const std::filesystem::path& tsv,
                             AmpliconSet& out) {
    PrimerMap primers;
    if (!read_primer_bed(bed, primers)) return false;
    if (tsv.empty()) {
        out = build_amplicon_set(std::move(primers), nullptr);
        return true;
    }
    std::vector<std::pair<std::string, std::string>> pairs;
    if (!read_primer_pairs_tsv(tsv, pairs)) return false;
    out = build_amplicon_set(std::move(primers), &pairs);
    return true;
}

}  // namespace bam_api
