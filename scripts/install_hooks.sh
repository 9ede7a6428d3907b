#!/bin/bash
# Installs the compile gate as the pre-commit hook: a commit whose native sources do not parse is refused.
cd "$(git rev-parse --show-toplevel)" || exit 1
printf '#!/bin/bash\nexec python3 scripts/check_tree.py --staged\n' > .git/hooks/pre-commit
chmod +x .git/hooks/pre-commit
echo "installed .git/hooks/pre-commit"
