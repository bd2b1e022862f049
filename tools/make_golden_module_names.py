"""Writes tests/golden/volumetric_module_names.json: every class, enum and function name the reference's `volumetric` extension
module binds, read from its binding sources (cpp/volumetric/volumetric_module.cpp and the *_module.h files it includes) - the
fixture tests/test_volumetric_module_names_cpu.py holds pyslam_amd.volumetric_module to.  Run where /root/reference exists."""
import json
import os
import re

REF = "/root/reference/cpp/volumetric"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "volumetric_module_names.json")


def main():
    names = set()
    for f in sorted(os.listdir(REF)):
        if not (f.endswith("_module.h") or f == "volumetric_module.cpp"):
            continue
        text = open(os.path.join(REF, f)).read()
        text = re.sub(r"\s+", " ", text)
        names.update(re.findall(r'py::(?:class_|enum_)<[^;]*?>\s*\(\s*m,\s*"(\w+)"', text))
        names.update(re.findall(r'm\.def\(\s*"(\w+)"', text))
        # macro-bound grids: DEFINE_*_BINDINGS(volumetric::Type, "Name")
        names.update(re.findall(r'DEFINE_\w+_BINDINGS\(\s*[\w:]+\s*,\s*"(\w+)"\s*\)', text))
        # the suffixed families of voxel_grid_data_module.h: std::string("Base") + suffix, bound for "" and "F"
        bases = re.findall(r'std::string\("(\w+)"\)\s*\+\s*suffix', text)
        for suffix in re.findall(r'bind_voxel_grid_data_family<[^>]*>\(m,\s*"(\w*)"\)', text):
            names.update(b + suffix for b in bases)
    names.discard("NAME")
    json.dump(sorted(names), open(OUT, "w"), indent=0)
    print(len(names), "names ->", OUT)


if __name__ == "__main__":
    main()
