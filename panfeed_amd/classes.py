"""Input records of the hot path (same field names and order as the reference's
/root/reference/panfeed/classes.py:5-18, so reference-shaped drivers can build them)."""
from collections import namedtuple

Feature = namedtuple("Feature", ["id", "chromosome", "start", "end", "strand"])

# sequence / compsequence: upper-case str of equal length; compsequence is the complement,
# NOT reversed, so compsequence[pos:pos+k][::-1] is the reverse complement of the window.
Seqinfo = namedtuple("Seqinfo", ["sequence", "compsequence", "id", "chromosome",
                                 "start", "end", "strand", "offset"])
