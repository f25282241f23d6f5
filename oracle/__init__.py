"""ORACLE package marker (test infrastructure). See oracle/oracle.h for what this is and who may use it."""
