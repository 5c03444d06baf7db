/* placeholder translation unit; the MC64 job-5 restatement lands here. */
int orc_mc64_placeholder(void) { return 0; }
