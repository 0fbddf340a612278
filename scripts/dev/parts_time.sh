#!/bin/bash
# time the parts kernel against the older plans on the same shapes
SH="${SH:-512,512,700,hinge 64,512,700,hinge 256,1000,220,dcg_hinge 32,1000,220,dcg_hinge 256,1000,220,logistic 128,600,136,hinge 512,512,700,logistic}"
echo "== parts wpc2"; LTR_PARTS_WPC=2 python scripts/dev/parts_check.py --time --nocheck --shapes $SH 2>&1 | grep shape
echo "== parts (auto)"; python scripts/dev/parts_check.py --time --nocheck --shapes $SH 2>&1 | grep shape
echo "== parts wpc3"; LTR_PARTS_WPC=3 python scripts/dev/parts_check.py --time --nocheck --shapes $SH 2>&1 | grep shape
echo "== parts wpc4"; LTR_PARTS_WPC=4 python scripts/dev/parts_check.py --time --nocheck --shapes $SH 2>&1 | grep shape
[ -n "$NOOLD" ] || { echo "== old plans"; LTR_DISABLE_PARTS=1 python scripts/dev/parts_check.py --time --nocheck --shapes $SH 2>&1 | grep shape; }
