"""Per-kernel table out of a rocprofv3 results .db (the default sqlite output): python tools/probe/db_stats.py file.db [n]"""
import re
import sqlite3
import sys
db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
rows = db.execute("select name, count(*), avg(end-start), min(end-start), sum(end-start), grid_x, workgroup_x, max(vgpr_count), max(accum_vgpr_count), max(lds_size) "
                  "from kernels group by name, grid_x order by sum(end-start) desc").fetchall()
for r in rows[:n]:
    nm = re.sub(r"void recon::\(anonymous namespace\)::|recon::\(anonymous namespace\)::|void ", "", r[0])[:78]
    print("%-80s n=%4d avg=%9.1f min=%9.1f tot=%10.1f us grid=%d wg=%d vgpr=%s agpr=%s lds=%s" % (nm, r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5], r[6], r[7], r[8], r[9]))
